"""The native step planner on the vector-observation agents' per-env loop (VERDICT r4 missing #4;
reference pfrl/agents/soft_actor_critic.py:354-374, td3.py:283-303): appends, queue bookkeeping
and the ``sample_n_k`` walk of an env range from ONE planner call and ONE transfer
(pfrl_amd/agents/_vector_device_step.py) must leave everything -- replay tables in HBM, host
mirrors, NumPy's stream, every action, loss and parameter -- exactly where the Python loop
leaves it.  (That loop is in turn held to the reference by the agent traces of
tests/test_agent_parity.py, which run through the planner as well.)"""
import tempfile

import numpy as np
import pytest
import torch


def _run(kind, planner_on, monkeypatch, n_envs=12, steps=720):
    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.agents import _vector_device_step as vds
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv
    from pfrl_amd.nn import ConcatObsAndAction, Lambda
    from test_agent_parity import _NoNoise, _squashed_head

    monkeypatch.setenv("PFRL_VECTOR_PLANNER", "1" if planner_on else "0")
    calls = {"native": 0, "python": 0, "updates": 0}
    orig = vds.plan_range

    def counting(*a, **kw):
        r = orig(*a, **kw)
        calls["native" if r is not None else "python"] += 1
        if r is not None:
            calls["updates"] += r[0]
        return r

    monkeypatch.setattr(vds, "plan_range", counting)
    obs_dim, act_dim = 24, 3
    pfrl.utils.set_random_seed(0)
    env = HostSyntheticVectorObsEnv(n_envs, obs_dim=obs_dim, act_dim=act_dim, seed=7, p_done=0.1)
    torch.manual_seed(97)

    def q():
        return torch.nn.Sequential(ConcatObsAndAction(), torch.nn.Linear(obs_dim + act_dim, 32),
                                   torch.nn.ReLU(), torch.nn.Linear(32, 1))

    burnin = lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32)   # noqa: E731
    rbuf = replay_buffers.ReplayBuffer(300)          # (the queue wraps: 720 appends)
    if kind == "sac":
        policy = torch.nn.Sequential(torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(),
                                     torch.nn.Linear(32, act_dim * 2), Lambda(_squashed_head))
        q1, q2 = q(), q()
        opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
        ag = agents.SoftActorCritic(policy, q1, q2, opts[0], opts[1], opts[2], rbuf, gamma=0.99,
                                    gpu=0, replay_start_size=100, minibatch_size=16,
                                    update_interval=2, burnin_action_func=burnin,
                                    entropy_target=None, initial_temperature=0.2,
                                    soft_update_tau=5e-3)
        nets = (policy, q1, q2, ag.target_q_func1)
    else:
        policy = torch.nn.Sequential(
            torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(), torch.nn.Linear(32, act_dim),
            pfrl.nn.BoundByTanh(low=-np.ones(act_dim, dtype=np.float32),
                                high=np.ones(act_dim, dtype=np.float32)),
            pfrl.policies.DeterministicHead())
        q1, q2 = q(), q()
        opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
        ag = agents.TD3(policy, q1, q2, opts[0], opts[1], opts[2], rbuf, gamma=0.99,
                        explorer=explorers.AdditiveGaussian(scale=0.1, low=-1.0, high=1.0),
                        gpu=0, replay_start_size=100, minibatch_size=16, update_interval=3,
                        soft_update_tau=5e-3, burnin_action_func=burnin, policy_update_delay=2)
        nets = (policy, q1, q2, ag.target_q_func1)
    actions = []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append(np.asarray(a, dtype=np.float32))
        return a

    ag.batch_act = spy_act
    with _NoNoise():
        pfrl.experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    torch.cuda.synchronize()
    st = rbuf.store
    st.flush()
    torch.cuda.synchronize()
    flat = lambda m: np.concatenate([p.detach().cpu().numpy().ravel() for p in m.parameters()])  # noqa: E731
    monkeypatch.setattr(vds, "plan_range", orig)
    out = dict(actions=np.asarray(actions), calls=dict(calls), agent=ag,
               np_state=np.random.get_state(), params=[flat(m) for m in nets],
               head=rbuf.memory.head, n_trans=st.n_trans, n_entries=st.n_entries,
               next_frame=st.frames.next_seq,
               stats=np.asarray([float(v) for _, v in ag.get_statistics()]))
    for name in ("t_state_ref", "t_next_ref", "t_action", "t_reward", "t_terminal", "e_tids", "e_len"):
        out[name] = getattr(st, name).cpu().numpy()
    out["frames"] = st.frames.frames.cpu().numpy()
    for name in ("h_state_ref", "h_next_ref", "h_action", "h_reward", "h_terminal", "h_min_fseq",
                 "h_e_tids", "h_e_len", "h_e_min_fseq"):
        out[name] = getattr(st, name).copy()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sac", "td3"])
def test_native_range_planner_equals_the_python_loop(kind, monkeypatch):
    want = _run(kind, False, monkeypatch)
    got = _run(kind, True, monkeypatch)
    assert want["calls"]["native"] == 0 and want["calls"]["python"] > 0
    # every range but the very first (whose appends create the tables) went through the planner,
    # and the planner drew every index set
    c = got["calls"]
    assert c["python"] <= 2 and c["native"] >= 100 and c["updates"] > 150, c
    assert got["agent"]._vec_planner is not None
    # NumPy's global stream: same position, same state
    assert got["np_state"][2] == want["np_state"][2]
    assert np.array_equal(got["np_state"][1], want["np_state"][1])
    for key in ("head", "n_trans", "n_entries", "next_frame"):
        assert got[key] == want[key], key
    for key in ("actions", "stats", "t_state_ref", "t_next_ref", "t_action", "t_reward",
                "t_terminal", "e_tids", "e_len", "frames", "h_state_ref", "h_next_ref", "h_action",
                "h_reward", "h_terminal", "h_min_fseq", "h_e_tids", "h_e_len", "h_e_min_fseq"):
        assert np.array_equal(got[key], want[key]), key
    for a, b in zip(got["params"], want["params"]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_ingest_vectors_equals_per_observation_ingest():
    """One pass over a list of f32 vectors == ``ingest`` one by one: same ring slots, same
    sequence numbers, identity hits where the per-observation cache would hit, same frames in
    HBM; lists the pass cannot take (another dtype of store, wrong size) are refused untouched."""
    from pfrl_amd.replay_buffers.device_replay import DeviceReplayStore

    rs = np.random.RandomState(3)
    obs = [rs.randn(24).astype(np.float32) for _ in range(10)]
    obs64 = [rs.randn(24) for _ in range(3)]
    a = DeviceReplayStore("cuda:0", 64, 1)
    b = DeviceReplayStore("cuda:0", 64, 1)
    assert b.ingest_vectors(obs[:2]) is None            # no store yet: the first ingest makes it
    for s in (a, b):
        s.ingest(obs[0])
    seq = [obs[1], obs[2], obs[1], obs[0], obs[3], obs64[0], obs[4], obs64[0]]
    want = [a.ingest(o) for o in seq]
    refs, seqs = b.ingest_vectors(seq)
    assert np.array_equal(refs[:, 0], [int(w[0][0]) for w in want])
    assert np.array_equal(seqs, [int(w[1]) for w in want])
    assert b.ingest_vectors([obs[5], np.zeros(7, dtype=np.float32)]) is None
    assert b.frames.next_seq == a.frames.next_seq
    # later per-observation ingests find what the pass stored
    assert int(b.ingest(obs[4])[0][0]) == int(a.ingest(obs[4])[0][0])
    a.flush()
    b.flush()
    torch.cuda.synchronize()
    n = a.frames.next_seq
    assert torch.equal(a.frames.frames[:n], b.frames.frames[:n])
