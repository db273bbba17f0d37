"""Replay-buffer API tests.

CPU: the host (gpu=None) back-end against the golden n-step / FIFO traces and
the reference's known-answer cases (tests/replay_buffers_test/
test_replay_buffer.py in the reference: n-step windows :69-144, capacity
:451-481, env_id isolation :639-702, batch_experiences :803-864).
GPU: the same traces through the HBM-resident back-end.
"""
import glob
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "replay_trace_*.npz")))


def _replay_trace(path, device):
    from pfrl_amd import replay_buffers
    from pfrl_amd.replay_buffer import batch_experiences

    g = np.load(path)
    seed, cap, n_steps, n_envs, batch = (int(x) for x in g["meta"])
    gamma = float(g["gamma"])
    np.random.seed(seed)
    rs = np.random.RandomState(seed + 5)
    obs_dim = 5
    n_ops = len(g["op_kind"])
    obs_table = rs.randn(n_ops + 2, obs_dim).astype(np.float32)
    rbuf = replay_buffers.ReplayBuffer(None if cap < 0 else cap, num_steps=n_steps,
                                       device=device, max_size=4096)
    tid = 0
    isample = 0
    dev = torch.device(device) if device else torch.device("cpu")
    for k, (kind, a, b) in enumerate(zip(g["op_kind"], g["op_a"], g["op_b"])):
        if kind == 1:
            rbuf.stop_current_episode(env_id=int(a))
        else:
            rbuf.append(state=obs_table[tid], action=int(g["action"][tid]),
                        reward=float(g["reward"][tid]), next_state=obs_table[tid + 1],
                        is_state_terminal=bool(b), env_id=int(a), tid=tid)
            tid += 1
        assert len(rbuf) == g["length"][k], k
        if isample < len(g["s_at_op"]) and g["s_at_op"][isample] == k:
            exps = rbuf.sample(batch)   # consumes np.random exactly like the reference
            sl = slice(isample * batch, (isample + 1) * batch)
            got_tids = [[t["tid"] for t in e] for e in exps]
            want = g["s_entry_tids"].reshape(-1, n_steps)[sl]
            assert got_tids == [[int(x) for x in row if x >= 0] for row in want]
            be = batch_experiences(exps, dev, lambda x: x, gamma)
            np.testing.assert_array_equal(be["reward"].cpu().numpy(), g["s_reward"][sl])
            np.testing.assert_array_equal(be["is_state_terminal"].cpu().numpy(),
                                          g["s_terminal"][sl])
            np.testing.assert_array_equal(be["discount"].cpu().numpy(), g["s_discount"][sl])
            np.testing.assert_array_equal(be["action"].cpu().numpy(), g["s_action"][sl])
            np.testing.assert_array_equal(be["state"].cpu().numpy(),
                                          obs_table[g["s_state_tid"][sl]])
            np.testing.assert_array_equal(be["next_state"].cpu().numpy(),
                                          obs_table[g["s_next_state_tid"][sl]])
            isample += 1
    final = [[t["tid"] for t in e] for e in rbuf.memory]
    assert [len(e) for e in final] == list(g["final_len"])
    assert final == [[int(x) for x in row if x >= 0] for row in g["final_tids"]]


@pytest.mark.parametrize("path", TRACES, ids=os.path.basename)
def test_uniform_replay_trace_host(path):
    _replay_trace(path, None)


@pytest.mark.gpu
@pytest.mark.parametrize("path", TRACES, ids=os.path.basename)
def test_uniform_replay_trace_device(path):
    _replay_trace(path, "cuda:0")


def _known_answer_batch_experiences(device):
    """reference tests/replay_buffers_test/test_replay_buffer.py:803-864"""
    from pfrl_amd import replay_buffers
    from pfrl_amd.replay_buffer import batch_experiences

    rbuf = replay_buffers.ReplayBuffer(100, num_steps=3, device=device)
    mk = lambda i, term=False: dict(state=np.float32([i]), action=i, reward=float(i),
                                    next_state=np.float32([i + 1]), is_state_terminal=term)
    for i in range(3):
        rbuf.append(**mk(i, term=(i == 2)))
    rbuf.append(**mk(10))
    rbuf.stop_current_episode()
    # memory: [0,1,2(term)], [1,2], [2], [10]
    assert len(rbuf) == 4
    ents = [rbuf.memory[i] for i in range(4)]
    assert [len(e) for e in ents] == [3, 2, 1, 1]
    dev = torch.device(device) if device else torch.device("cpu")
    if device:
        from pfrl_amd.replay_buffer import DeviceExperienceBatch

        seqs = np.arange(4, dtype=np.int64)
        exps = DeviceExperienceBatch(rbuf.store, rbuf.store.slots_for(seqs), seqs)
    else:
        exps = ents
    be = batch_experiences(exps, dev, lambda x: x, 0.99)
    np.testing.assert_allclose(be["is_state_terminal"].cpu().numpy(), [1, 1, 1, 0])
    np.testing.assert_allclose(be["discount"].cpu().numpy(),
                               np.float32([0.99 ** 3, 0.99 ** 2, 0.99, 0.99]))
    np.testing.assert_allclose(be["reward"].cpu().numpy(),
                               np.float32([0 + 0.99 * 1 + 0.99 ** 2 * 2, 1 + 0.99 * 2, 2, 10]))
    np.testing.assert_array_equal(be["next_state"].cpu().numpy().ravel(), [3, 3, 3, 11])
    np.testing.assert_array_equal(be["state"].cpu().numpy().ravel(), [0, 1, 2, 10])


def test_batch_experiences_known_answer_host():
    _known_answer_batch_experiences(None)


@pytest.mark.gpu
def test_batch_experiences_known_answer_device():
    _known_answer_batch_experiences("cuda:0")


def _capacity_and_misuse(device):
    from pfrl_amd import replay_buffers

    rbuf = replay_buffers.ReplayBuffer(5, device=device)
    for i in range(12):
        rbuf.append(np.float32([i]), i, 1.0, np.float32([i + 1]))
        assert len(rbuf) == min(i + 1, 5)
    assert [e[0]["action"] for e in rbuf.memory] == [7, 8, 9, 10, 11]
    with pytest.raises(AssertionError):
        rbuf.sample(6)
    with pytest.raises(ValueError):
        from pfrl_amd.utils.random import sample_n_k

        sample_n_k(3, 4)


def test_capacity_and_misuse_host():
    _capacity_and_misuse(None)


@pytest.mark.gpu
def test_capacity_and_misuse_device():
    _capacity_and_misuse("cuda:0")


@pytest.mark.gpu
def test_device_replay_vector_obs_continuous_actions():
    """SAC-shaped data path (config 5): float32 observations [376], float32
    actions [17], B = 256; fused gather is a plain f32 copy."""
    import oracle
    from pfrl_amd import replay_buffers
    from pfrl_amd.replay_buffer import batch_experiences

    rs = np.random.RandomState(0)
    np.random.seed(0)
    n, B = 3000, 256
    obs = rs.randn(n + 1, 376).astype(np.float32)
    act = rs.uniform(-1, 1, size=(n, 17)).astype(np.float32)
    rew = rs.randn(n)
    term = rs.rand(n) < 0.001
    rbuf = replay_buffers.ReplayBuffer(2000, device="cuda:0")
    for i in range(n):
        rbuf.append(obs[i], act[i], float(rew[i]), obs[i + 1], is_state_terminal=bool(term[i]),
                    idx=i)
    exps = rbuf.sample(B)
    ids = [e[0]["idx"] for e in exps]
    be = batch_experiences(exps, torch.device("cuda:0"), lambda x: x, 0.99)
    np.testing.assert_array_equal(be["state"].cpu().numpy(), obs[ids])
    np.testing.assert_array_equal(be["next_state"].cpu().numpy(), obs[np.asarray(ids) + 1])
    np.testing.assert_array_equal(be["action"].cpu().numpy(), act[ids])
    want = oracle.batch_experiences_scalars([[i] for i in ids], rew, term, 0.99, 1)
    np.testing.assert_array_equal(be["reward"].cpu().numpy(), want["reward"])
    np.testing.assert_array_equal(be["is_state_terminal"].cpu().numpy(), want["is_state_terminal"])
    # all minibatches of a batched env step in one launch (SAC / TD3 step-fused path): 17 x 256
    # entries take the wave-per-frame kernel; every value against the host arrays / the oracle
    sets = [rbuf.lookahead_sample(B) for _ in range(17)]
    big = rbuf.fetch_many(sets, lambda x: x, 0.99)
    assert big["state"].shape == (17, B, 376)
    head = rbuf.memory.head
    for u, seqs in enumerate(sets):
        ids = [rbuf.memory[int(q - head)][0]["idx"] for q in seqs]
        np.testing.assert_array_equal(big["state"][u].cpu().numpy(), obs[ids])
        np.testing.assert_array_equal(big["next_state"][u].cpu().numpy(), obs[np.asarray(ids) + 1])
        np.testing.assert_array_equal(big["action"][u].cpu().numpy(), act[ids])
        want = oracle.batch_experiences_scalars([[i] for i in ids], rew, term, 0.99, 1)
        for key in ("reward", "is_state_terminal", "discount"):
            np.testing.assert_array_equal(big[key][u].cpu().numpy(), want[key])


@pytest.mark.gpu
def test_prioritized_replay_buffer_api_device():
    """PER wrapper behaviour the reference pins (test_replay_buffer.py:358,
    392-449): weight 1.0 for a single item; equal weights for equal (clipped)
    errors; 'memory' normalisation uses the global minimum priority."""
    from pfrl_amd import replay_buffers

    rbuf = replay_buffers.PrioritizedReplayBuffer(100, device="cuda:0", normalize_by_max="batch")
    rbuf.append(np.float32([0]), 0, 1.0, np.float32([1]))
    s = rbuf.sample(1)
    assert s[0][0]["weight"] == pytest.approx(1.0)
    rbuf.update_errors([3.14])
    for i in range(1, 8):
        rbuf.append(np.float32([i]), i, 1.0, np.float32([i + 1]))
    s = rbuf.sample(4)
    rbuf.update_errors([5.0, 7.0, 100.0, 1.5])   # all clipped to error_max=1
    s = rbuf.sample(4)
    ws = [e[0]["weight"] for e in s]
    rbuf.update_errors([1.0] * 4)
    assert len(rbuf) == 8
    with pytest.raises(AssertionError):
        rbuf.update_errors([1.0])   # no pending sample
    rb2 = replay_buffers.PrioritizedReplayBuffer(100, device="cuda:0", normalize_by_max="memory",
                                                 alpha=1.0, beta0=1.0, betasteps=None, eps=0.0,
                                                 error_min=None, error_max=None)
    for i in range(4):
        rb2.append(np.float32([i]), i, 1.0, np.float32([i + 1]))
    first = rb2.sample(4)
    errs = [1.0, 2.0, 4.0, 8.0]
    rb2.update_errors(errs)   # priorities go to the sampled items, in sampled order
    pri = {e[0]["action"]: p for e, p in zip(first, errs)}
    s = rb2.sample(2)
    for e in s:
        # w = (p/total / (min/total)) ** -1 = min / p
        assert e[0]["weight"] == pytest.approx(1.0 / pri[e[0]["action"]], rel=1e-6)
    rb2.update_errors([1.0, 1.0])


@pytest.mark.gpu
def test_host_lazyframes_ingest_dedupes_frames_and_matches_phi():
    """Real-env ingest: LazyFrames from VectorFrameStack are uploaded one frame
    at a time, shared frames exactly once, and batch_experiences reproduces
    phi(np.asarray(obs)) bit-for-bit (reference train_dqn_batch_ale.py:229-231)."""
    from pfrl_amd import replay_buffers
    from pfrl_amd.envs import SerialVectorEnv
    from pfrl_amd.replay_buffer import batch_experiences
    from pfrl_amd.wrappers import VectorFrameStack

    class Env:
        def __init__(self, seed):
            self.rs = np.random.RandomState(seed)

        def reset(self):
            return self.rs.randint(0, 256, size=(1, 84, 84)).astype(np.uint8)

        def step(self, a):
            return (self.rs.randint(0, 256, size=(1, 84, 84)).astype(np.uint8), 1.0,
                    bool(self.rs.rand() < 0.1), {})

        def close(self):
            pass

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    N, steps = 3, 40
    venv = VectorFrameStack(SerialVectorEnv([Env(i) for i in range(N)]), 4, stack_axis=0)
    rbuf = replay_buffers.ReplayBuffer(1000, device="cuda:0")
    rbuf.store.set_phi(phi)
    obs = venv.reset()
    host_log = []
    n_frames = N
    for t in range(steps):
        nobs, r, dones, _ = venv.step([0] * N)
        n_frames += N
        for i in range(N):
            rbuf.append(obs[i], 1, r[i], nobs[i], is_state_terminal=dones[i], env_id=i,
                        idx=len(host_log))
            host_log.append((np.asarray(obs[i]), np.asarray(nobs[i])))
        obs = venv.reset(np.logical_not(dones))
        n_frames += int(np.sum(dones))
    np.random.seed(0)
    exps = rbuf.sample(32)
    be = batch_experiences(exps, torch.device("cuda:0"), phi, 0.99)
    ids = [e[0]["idx"] for e in exps]
    np.testing.assert_array_equal(be["state"].cpu().numpy(),
                                  np.stack([phi(host_log[i][0]) for i in ids]))
    np.testing.assert_array_equal(be["next_state"].cpu().numpy(),
                                  np.stack([phi(host_log[i][1]) for i in ids]))
    assert be["state"].shape == (32, 4, 84, 84)
    # every distinct frame went to HBM exactly once
    assert rbuf.store.frames.next_seq == n_frames


@pytest.mark.gpu
@pytest.mark.parametrize("prioritized", [False, True])
def test_native_checkpoint_round_trip(tmp_path, prioritized):
    """save(native=True) -> load into a fresh buffer: same length, same sampled
    minibatches for the same NumPy stream, same priority tree."""
    from pfrl_amd import replay_buffers
    from pfrl_amd.replay_buffer import batch_experiences

    def make():
        if prioritized:
            return replay_buffers.PrioritizedReplayBuffer(300, num_steps=3, device="cuda:0")
        return replay_buffers.ReplayBuffer(300, num_steps=3, device="cuda:0")

    rs = np.random.RandomState(0)
    a = make()
    obs = [rs.randint(0, 256, size=(4, 12, 12)).astype(np.uint8) for _ in range(502)]
    for i in range(500):
        a.append(obs[i], int(rs.randint(4)), float(rs.randn()), obs[i + 1],
                 is_state_terminal=bool(rs.rand() < 0.05), env_id=i % 3)
        if prioritized and i > 50 and i % 7 == 0:
            a.sample(8)
            a.update_errors(torch.rand(8, device="cuda:0") * 1.5)
    path = str(tmp_path / "replay.pt")
    a.save(path, native=True)
    b = make()
    b.load(path)
    assert len(a) == len(b)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    a.store.set_phi(phi)
    b.store.set_phi(phi)
    for buf in (a, b):
        np.random.seed(123)
        buf._be = batch_experiences(buf.sample(16), torch.device("cuda:0"), phi, 0.99)
        buf._be = {k: v.clone() for k, v in buf._be.items()}
    for k in a._be:
        assert torch.equal(a._be[k], b._be[k]), k
    if prioritized:
        assert a.memory.tree.root_stats() == b.memory.tree.root_stats()
    # both keep working after the restore (window state was restored too)
    for buf in (a, b):
        if prioritized:
            buf.update_errors(torch.full((16,), 0.5, device="cuda:0"))
        buf.append(obs[500], 1, 1.0, obs[501], env_id=0)
    assert len(a) == len(b)


# ---------------------------------------------------------------------------------------------
# replay checkpoints the reference can read
# ---------------------------------------------------------------------------------------------
REFERENCE = os.environ.get("PFRL_REFERENCE", "/root/reference")
_needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "pfrl")),
                                      reason="the reference checkout exists in the build container only")


def _run_with_reference(code):
    """Run ``code`` in a fresh interpreter that can import the REFERENCE (and the test-only gym
    stand-in) but NOT pfrl_amd: what a user of the reference has installed."""
    import subprocess
    import sys

    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([REFERENCE, os.path.join(os.path.dirname(__file__), "_gymshim")])
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@_needs_reference
def test_saved_replay_is_read_by_the_reference_and_back(tmp_path):
    """ReplayBuffer.save writes a plain deque of dict lists with NumPy observations: the
    reference's load (pfrl/replay_buffers/replay_buffer.py:89-94) reads it with no pfrl_amd
    class on its path; and a file the reference wrote loads here with no pfrl installed."""
    from pfrl_amd import replay_buffers
    from pfrl_amd.wrappers.atari_wrappers import LazyFrames

    rs = np.random.RandomState(0)
    rbuf = replay_buffers.ReplayBuffer(50, num_steps=2)
    frames = [rs.randint(0, 256, size=(1, 6, 6)).astype(np.uint8) for _ in range(40)]
    for i in range(30):
        s = LazyFrames(frames[i:i + 4], stack_axis=0)
        ns = LazyFrames(frames[i + 1:i + 5], stack_axis=0)
        rbuf.append(s, i % 3, float(i), ns, is_state_terminal=(i % 7 == 6))
        if i % 7 == 6:
            rbuf.stop_current_episode()
    # default save: the entries as they are -- sharing intact, as in the reference's own save
    shared = str(tmp_path / "shared.pkl")
    rbuf.save(shared)
    again = replay_buffers.ReplayBuffer(50, num_steps=2)
    again.load(shared)
    assert len(again) == len(rbuf)
    e0, e1 = again.memory[0], again.memory[1]
    assert e0[1] is e1[0]                                       # one dict per transition
    assert e0[0]["next_state"] is e0[1]["state"] or \
        e0[0]["next_state"]._frames[0] is e0[1]["state"]._frames[0]   # one array per frame
    assert os.path.getsize(shared) < 40 * 36 * 3 + 30 * 600     # ~ frames once + scalars
    ours = str(tmp_path / "ours.pkl")
    rbuf.save(ours, materialize=True)
    assert b"pfrl_amd" not in open(ours, "rb").read()
    theirs = str(tmp_path / "theirs.pkl")
    out = _run_with_reference(
        "import sys, numpy as np\n"
        "assert not any('pfrl_amd' in m for m in sys.modules)\n"
        "import pfrl\n"
        "rb = pfrl.replay_buffers.ReplayBuffer(50, num_steps=2)\n"
        "rb.load(%r)\n"
        "assert type(rb.memory).__module__ == 'pfrl.collections.random_access_queue'\n"
        "ents = [rb.memory[i] for i in range(len(rb))]\n"
        "print(len(rb), sum(len(e) for e in ents), float(sum(t['reward'] for e in ents for t in e)),"
        " int(sum(int(np.asarray(e[0]['state']).sum()) for e in ents)))\n"
        "np.random.seed(1); be = pfrl.replay_buffer.batch_experiences(rb.sample(4), 'cpu', lambda x: x, 0.9)\n"
        "assert be['state'].shape == (4, 4, 6, 6)\n"
        "rb.save(%r)\n" % (ours, theirs))
    n, n_trans, rsum, ssum = out.split()
    ents = [rbuf.memory[i] for i in range(len(rbuf))]
    assert int(n) == len(rbuf) and int(n_trans) == sum(len(e) for e in ents)
    assert float(rsum) == sum(t["reward"] for e in ents for t in e)
    assert int(ssum) == sum(int(np.asarray(e[0]["state"]).sum()) for e in ents)
    # and back: the reference's own pickle (its RandomAccessQueue class) into this package
    assert b"pfrl.collections.random_access_queue" in open(theirs, "rb").read()
    back = replay_buffers.ReplayBuffer(50, num_steps=2)
    back.load(theirs)
    assert len(back) == len(rbuf)
    for a, b in zip(back.memory, rbuf.memory):
        assert len(a) == len(b)
        for ta, tb in zip(a, b):
            assert ta["reward"] == tb["reward"] and ta["action"] == tb["action"]
            assert np.array_equal(np.asarray(ta["state"]), np.asarray(tb["state"]))


@_needs_reference
def test_device_buffer_checkpoint_is_read_by_the_reference():
    """tests/golden/device_replay_save.pkl was written on the MI355X by ``save()`` of an
    HBM-resident buffer (make_device_replay_pickle.py); the reference reads it and finds the
    observations / scalars recorded next to it."""
    pkl = os.path.join(GOLDEN, "device_replay_save.pkl")
    if not os.path.exists(pkl):
        pytest.skip("fixture not generated yet (needs the GPU box)")
    g = np.load(os.path.join(GOLDEN, "device_replay_save.npz"))
    assert b"pfrl_amd" not in open(pkl, "rb").read()
    out = _run_with_reference(
        "import numpy as np, pfrl, json\n"
        "rb = pfrl.replay_buffers.ReplayBuffer(40, num_steps=3)\n"
        "rb.load(%r)\n"
        "ents = [rb.memory[i] for i in range(len(rb))]\n"
        "np.savez('/tmp/_ref_read.npz', lens=np.array([len(e) for e in ents]),"
        " first_state=np.stack([np.asarray(e[0]['state']) for e in ents]),"
        " last_next_state=np.stack([np.asarray(e[-1]['next_state']) for e in ents]),"
        " actions=np.array([e[0]['action'] for e in ents]),"
        " terminal=np.array([e[-1]['is_state_terminal'] for e in ents]))\n"
        "np.random.seed(0); be = pfrl.replay_buffer.batch_experiences(rb.sample(8), 'cpu',"
        " lambda x: np.asarray(x, dtype=np.float32) / 255, 0.99)\n"
        "print(tuple(be['state'].shape))\n" % pkl)
    assert out.strip() == "(8, 4, 12, 12)"
    r = np.load("/tmp/_ref_read.npz")
    for key in ("lens", "first_state", "last_next_state", "actions", "terminal"):
        assert np.array_equal(r[key], g[key]), key


@pytest.mark.gpu
def test_device_buffer_save_is_portable_and_dqn_snapshot_round_trips(tmp_path):
    """save() of an HBM-resident buffer names no pfrl_amd class; a host buffer loads it;
    DQN.save_snapshot / load_snapshot (reference pfrl/agents/dqn.py:794-810) restore t, optim_t,
    the replay contents and the networks on the device path, and training continues."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.q_functions import DiscreteActionValueHead

    dev = torch.device("cuda:0")

    def make():
        pfrl.utils.set_random_seed(0)
        store = DeviceFrameStore(2048, (12, 12), torch.uint8, dev, stack=4)
        env = SyntheticAtariVectorEnv(4, store=store, seed=3, n_actions=4, p_done=0.05)
        torch.manual_seed(1)
        q = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU(),
                                torch.nn.Linear(32, 4), DiscreteActionValueHead())
        opt = torch.optim.RMSprop(q.parameters(), lr=1e-3)
        rbuf = replay_buffers.ReplayBuffer(300, num_steps=2)
        ex = explorers.ConstantEpsilonGreedy(0.3, lambda: np.random.randint(4))
        ag = agents.DQN(q, opt, rbuf, 0.99, ex, gpu=0, replay_start_size=32, minibatch_size=8,
                        update_interval=4, target_update_interval=40,
                        phi=lambda x: np.asarray(x, dtype=np.float32) / 255)
        return ag, env, q, rbuf

    ag, env, q, rbuf = make()
    pfrl.experiments.train_agent_batch(ag, env, 400, str(tmp_path / "o"))
    snap = str(tmp_path / "snap")
    os.makedirs(snap)
    ag.save_snapshot(snap)
    raw = open(os.path.join(snap, "replay_buffer.pkl"), "rb").read()
    assert b"pfrl_amd" not in raw
    host = replay_buffers.ReplayBuffer(300, num_steps=2)
    host.load(os.path.join(snap, "replay_buffer.pkl"))
    assert len(host) == len(rbuf)
    for a, b in zip(host.memory, rbuf.memory):
        assert [t["reward"] for t in a] == [t["reward"] for t in b]
        assert np.array_equal(np.asarray(a[0]["state"]), np.asarray(b[0]["state"]))
    ag2, env2, q2, rbuf2 = make()
    ag2.load_snapshot(snap)
    assert (ag2.t, ag2.optim_t, ag2.cumulative_steps) == (ag.t, ag.optim_t, ag.cumulative_steps)
    assert len(rbuf2) == len(rbuf) and rbuf2.is_device
    for (k, v), (_, w) in zip(q2.state_dict().items(), q.state_dict().items()):
        assert torch.equal(v, w), k
    for a, b in zip(rbuf2.memory, rbuf.memory):
        assert [t["action"] for t in a] == [t["action"] for t in b]
        assert np.array_equal(np.asarray(a[-1]["next_state"]), np.asarray(b[-1]["next_state"]))
    n0 = ag2.optim_t
    pfrl.experiments.train_agent_batch(ag2, env2, 80, str(tmp_path / "o2"), step_offset=ag2.t)
    assert ag2.optim_t > n0
