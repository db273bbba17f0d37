"""Parity of the EXACT paths ``bench.py`` times, at their own shapes, against the oracle.

The agent-trace tests (test_agent_parity.py) compare small host-env runs with traces recorded
from the reference.  These tests pin what the benchmark runs: the device env
(``SyntheticAtariVectorEnv`` writing 84x84 u8 frames into the HBM ring), 256 / 512 envs, the
Nature CNN in channels_last, the step-fused gather in two env ranges, rings small enough to
wrap the entry ring, the frame ring and the ``slack`` rows several times.  The expectation is
built independently of the product code: a Python deque of what the reference's loop would
append (pfrl/agents/dqn.py:509-549), index sets drawn by a restatement of
pfrl/utils/random.py:4-28 from the same NumPy stream position, and minibatches computed by the
C oracle (oracle/pfrl_oracle.c) from a D2H copy of the frame ring.  Everything integer / byte
is compared bit-exact.
"""
import argparse
import collections
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _bench_args(**kw):
    d = dict(gpus=1, algo="dqn", host_env=False, steps=1, warmup=0, num_envs=256, capacity=10 ** 6,
             minibatch=32, update_interval=4, seed=0, prefill=None, no_cpu_baseline=True,
             cpu_baseline_seconds=0.0, cpu_baseline_threads=1, cudnn_benchmark=False,
             channels_last=True, blas="default", chunks=None, torch_optimizer=False,
             replay_start=None, frame_slots=None, slack=None, priority_pow="device")
    d.update(kw)
    return argparse.Namespace(**d)


def _ref_sample_n_k(n, k):
    """k distinct uniform indices from range(n), consuming the legacy global NumPy stream the
    way pfrl/utils/random.py:4-28 does (choice without replacement when 3k >= n; otherwise 2k
    draws with replacement, first-come acceptance, refill of the spare half when exhausted)."""
    if 3 * k >= n:
        return np.random.choice(n, k, replace=False)
    res = np.random.choice(n, 2 * k)
    seen = set()
    spare = k
    for i in range(k):
        x = res[i]
        while x in seen:
            x = res[i] = res[spare]
            spare += 1
            if spare == 2 * k:
                res[k:] = np.random.choice(n, k)
                spare = k
        seen.add(x)
    return res[:k]


def _same_rng_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_dqn_bench_path_minibatches_match_oracle():
    """BASELINE configs[1] as bench.py builds it (256 envs, 84x84x4, Nature CNN channels_last,
    step_fused_chunks=(0.1, 0.4), B=32, update_interval=4): for checked steps after every ring
    has wrapped, all 64 minibatches of the step -- index sets, action / reward / terminal /
    discount and the gathered fp32 state / next_state stacks -- equal the oracle bit for bit,
    the NumPy stream ends at the same position, and the first TD loss of the step agrees with
    the same network evaluated by stock PyTorch on the CPU to 1e-5."""
    import bench
    import oracle

    dev = torch.device("cuda:0")
    N, B, CAP, START = 256, 32, 1500, 1024
    args = _bench_args(capacity=CAP, frame_slots=6144, slack=512, replay_start=START)
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    assert agent.step_fused_chunks == (0.1, 0.4) and agent.step_fused_gather and agent.use_graphs
    gamma = agent.gamma

    fetched, keep = [], [False]
    orig_fetch = rbuf.fetch_many

    def spy_fetch(seq_sets, phi, g):
        big = orig_fetch(seq_sets, phi, g)
        if keep[0]:
            fetched.append({k: v.detach().cpu().numpy() for k, v in big.items()
                            if k != "target_next_raw"})
        return big

    rbuf.fetch_many = spy_fetch
    # the native step (agents/_dqn_device_step.py) gathers from slots planned in C
    orig_fetch_slots = rbuf.store.fetch_many_slots
    native_calls = [0]

    def spy_fetch_slots(slots_dev, U, B_, phi, g):
        big = orig_fetch_slots(slots_dev, U, B_, phi, g)
        native_calls[0] += 1
        if keep[0]:
            fetched.append({k: v.detach().cpu().numpy() for k, v in big.items()
                            if k != "target_next_raw"})
        return big

    rbuf.store.fetch_many_slots = spy_fetch_slots
    losses = []
    orig_extend = agent.loss_record.extend

    def spy_extend(t):
        orig_extend(t)
        losses.append(t.detach().float().reshape(-1).cpu().numpy())

    agent.loss_record.extend = spy_extend

    model = collections.deque(maxlen=CAP)     # what pfrl's ReplayBuffer(CAP).memory would hold
    t_counter = 0
    checked = {30: 0, 52: 0, 71: 0}
    obss = env.reset()
    frames_written0 = env.store.next_seq
    for step in range(72):
        state_refs = obss.refs.copy()
        actions = np.asarray(agent.batch_act(obss))
        obss2, rs, dones, _ = env.step(actions)
        next_refs = obss2.refs.copy()
        keep[0] = step in checked
        del fetched[:]
        del losses[:]
        if keep[0]:
            cpu_q = copy.deepcopy(agent.model).cpu().to(memory_format=torch.contiguous_format)
            cpu_t = copy.deepcopy(agent.target_model).cpu().to(memory_format=torch.contiguous_format)
        s0 = np.random.get_state()
        agent.batch_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
        s1 = np.random.get_state()
        # the reference's loop, on the model (pfrl/agents/dqn.py:516-549)
        np.random.set_state(s0)
        expected = []
        for i in range(N):
            t_counter += 1
            model.append((state_refs[i], int(actions[i]), float(rs[i]), next_refs[i], bool(dones[i])))
            if len(model) >= START and t_counter % 4 == 0:
                expected.append([model[j] for j in _ref_sample_n_k(len(model), B)])
        assert _same_rng_state(np.random.get_state(), s1), "NumPy stream position differs"
        assert agent.t == t_counter and len(rbuf) == len(model)
        if keep[0]:
            assert len(expected) == 64
            got = {k: np.concatenate([f[k] for f in fetched]) for k in fetched[0]}
            assert got["state"].shape == (64, B, 4, 84, 84)
            torch.cuda.synchronize()
            frames = env.store.frames.cpu().numpy().reshape(env.store.n_slots, -1)
            for u, ents in enumerate(expected):
                want_s = oracle.batch_states_u8(frames, np.stack([e[0] for e in ents]), 255.0)
                want_n = oracle.batch_states_u8(frames, np.stack([e[3] for e in ents]), 255.0)
                assert np.array_equal(got["state"][u].reshape(B, -1), want_s.reshape(B, -1)), (step, u)
                assert np.array_equal(got["next_state"][u].reshape(B, -1), want_n.reshape(B, -1)), (step, u)
                sc = oracle.batch_experiences_scalars(
                    [[b] for b in range(B)], np.array([e[2] for e in ents]),
                    np.array([e[4] for e in ents], dtype=np.uint8), gamma, 1)
                assert np.array_equal(got["action"][u], np.array([e[1] for e in ents]))
                for key in ("reward", "is_state_terminal", "discount"):
                    assert np.array_equal(got[key][u], sc[key]), (step, u, key)
            # TD loss of the step's first update, same weights, stock PyTorch on the CPU
            e0 = expected[0]
            s = torch.from_numpy(oracle.batch_states_u8(frames, np.stack([e[0] for e in e0]), 255.0)
                                 ).view(B, 4, 84, 84)
            ns = torch.from_numpy(oracle.batch_states_u8(frames, np.stack([e[3] for e in e0]), 255.0)
                                  ).view(B, 4, 84, 84)
            with torch.no_grad():
                y = cpu_q(s).evaluate_actions(torch.tensor([e[1] for e in e0]))
                nq = cpu_t(ns).max
                r = torch.tensor([e[2] for e in e0], dtype=torch.float32)
                term = torch.tensor([float(e[4]) for e in e0])
                tgt = r + (gamma * (1.0 - term)) * nq
                want_loss = float(torch.nn.functional.smooth_l1_loss(y, tgt, reduction="sum"))
            got_loss = float(np.concatenate(losses)[0])
            assert abs(got_loss - want_loss) <= 1e-5 * max(1.0, abs(want_loss)), (got_loss, want_loss)
            checked[step] = 1
        obss = env.reset(~np.asarray(dones))
    assert all(checked.values())
    st = rbuf.store
    # every ring wrapped: entries / transitions (E = R = CAP + slack), frames
    assert st.n_entries > 3 * st.E and st.n_trans > 3 * st.R
    assert env.store.next_seq - frames_written0 > 2 * env.store.n_slots
    assert any(k[0] == "range" for k in agent._graphed.graphs)   # the path bench.py times
    # ... including the native step: device-resident actions, planner-drawn index sets
    from pfrl_amd.device_store import DeviceActions

    assert agent.device_step and native_calls[0] >= 60      # one planned range per updating step
    assert isinstance(agent.batch_last_action.batch, DeviceActions)


@pytest.mark.parametrize("priority_pow", ["device", "host_libm"])
def test_rainbow_bench_path_tree_matches_oracle(priority_pow):
    """BASELINE configs[2] as bench.py builds it (CategoricalDoubleDQN, 256 envs, PER n = 3,
    normalize_by_max='memory', replay stream on): every sample's indices and importance
    weights and the tree's root sum / min / max_priority after every update_errors equal
    ``OraclePrioritizedBuffer`` fed the same appends, the same uniform draws and the device's
    TD errors -- bit for bit in the mode ``bench.py --algo rainbow`` runs ('device': glibc's
    powf restated in the update kernel, csrc/powf_glibc.h) and in 'host_libm'."""
    import bench
    import oracle
    from pfrl_amd.collections import prioritized as dev_pri

    dev = torch.device("cuda:0")
    N, CAP, START = 256, 3000, 1024
    args = _bench_args(algo="rainbow", capacity=CAP, frame_slots=3000 + 24 * N + 512, slack=1024,
                       replay_start=START, priority_pow=priority_pow)
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    assert agent._replay_stream is not None
    assert rbuf.priority_pow == priority_pow
    if priority_pow == "device":
        # ... which is what `bench.py --algo rainbow` runs when no flag is given
        saved, sys.argv = sys.argv, ["bench.py", "--algo", "rainbow"]
        try:
            assert bench.parse_args().priority_pow == "device"
        finally:
            sys.argv = saved
    orc = oracle.OraclePrioritizedBuffer(CAP)
    counts = dict(samples=0, updates=0, appends=0)
    pending = {}

    orig_append = dev_pri.PrioritizedBuffer.append
    orig_sample = dev_pri.PrioritizedBuffer.sample_device

    # The agent keeps the host one update point ahead of the device (DQN._batch_observe_train_per):
    # the appends up to the next sample point and that sample's preparation are CALLED before the
    # update that sets the previous minibatch's priorities is launched.  The oracle is fed in the
    # reference's logical order: appends made while a sample is waiting for its priorities are
    # replayed right after them, and a prepared sample is drawn when it is finished.
    late_appends = []

    def spy_append(self, value, priority=None):
        assert priority is None
        if self.flag_wait_priority:
            late_appends.append(value)
        else:
            orc.append(value)
        counts["appends"] += 1
        return orig_append(self, value, priority)

    def check_sample(self, out, want, beta):
        self._join()
        x = out["x"].cpu().numpy()
        np.testing.assert_array_equal(x - self.frame.head, want["indices"])
        np.testing.assert_array_equal(out["pri"].cpu().numpy(), want["priorities"])
        # reference weights (pfrl/replay_buffers/prioritized.py:57-66, normalize_by_max="memory")
        w = (want["probabilities"] / want["min_prob"]) ** (-beta)
        np.testing.assert_allclose(out["weight"].cpu().numpy(), w, rtol=1e-5)
        counts["samples"] += 1
        pending["x"] = x

    def spy_sample(self, n, u01=None, normalize=1, beta=0.0, slot_mod=0, **kw):
        assert u01 is None
        u = np.random.random_sample(n)          # the draws np.random.uniform would consume
        if kw.get("split"):
            out, finish = orig_sample(self, n, u01=u, normalize=normalize, beta=beta,
                                      slot_mod=slot_mod, **kw)
            counts["split"] = counts.get("split", 0) + 1

            def finish_checked():
                want = orc.sample(u)
                r = finish()
                check_sample(self, out, want, beta)
                return r

            return out, finish_checked
        want = orc.sample(u)
        out = orig_sample(self, n, u01=u, normalize=normalize, beta=beta, slot_mod=slot_mod, **kw)
        check_sample(self, out, want, beta)
        return out

    dev_pri.PrioritizedBuffer.append = spy_append
    dev_pri.PrioritizedBuffer.sample_device = spy_sample
    orig_update = rbuf.update_errors

    def spy_update(errors):
        err = errors.detach().float().cpu().numpy().reshape(-1)
        v, t = oracle.priority_from_errors_f32(err, rbuf.error_min, rbuf.error_max, rbuf.eps,
                                               rbuf.alpha)
        orc.set_last_priority(v, t)
        orig_update(errors)
        tree = rbuf.memory.tree
        if not late_appends:
            # (with appends already recorded on the device side the root is not a state the
            # reference ever shows; the next sample's indices and priorities check it instead)
            got, so = tree.root_stats(), orc.stats()
            assert got[0] == so["sum"] and got[1] == so["min"] and got[2] == so["max_priority"], \
                (counts, got, so)
        for value in late_appends:
            orc.append(value)
        del late_appends[:]
        counts["updates"] += 1

    rbuf.update_errors = spy_update
    try:
        obss = env.reset()
        for step in range(16):
            obss = bench.one_step(agent, env, obss, N)
    finally:
        dev_pri.PrioritizedBuffer.append = orig_append
        dev_pri.PrioritizedBuffer.sample_device = orig_sample
    assert counts["updates"] >= 300 and counts["samples"] >= counts["updates"]
    assert counts["appends"] > CAP          # the tree frame slid / re-rooted past capacity
    assert len(rbuf) == len(orc) == CAP


def test_ppo_bench_path_gae_and_advantage_statistics_match_oracle(monkeypatch):
    """BASELINE configs[3] as bench.py builds it (512 envs x 128 steps): the rollout the agent
    hands to ``gae_scan`` / ``adv_stats`` gives, fragment by fragment, the oracle's advantages
    and value targets bit for bit, and its mean / std to 1e-6."""
    import bench
    import oracle
    from pfrl_amd import ops

    dev = torch.device("cuda:0")
    N, T = 512, 128
    args = _bench_args(algo="ppo", num_envs=N)
    agent, env, _ = bench.build_agent(args, dev, 0)
    seen = {}
    orig_gae, orig_stats = ops.gae_scan, ops.adv_stats

    def spy_gae(reward, v_pred, next_v, nonterm, cut, gamma, lambd, mode=0):
        adv, vt = orig_gae(reward, v_pred, next_v, nonterm, cut, gamma, lambd, mode)
        seen["gae"] = [x.detach().cpu().numpy() for x in (reward, v_pred, next_v, nonterm, cut, adv, vt)]
        seen["gae_args"] = (gamma, lambd, mode)
        return adv, vt

    def spy_stats(adv):
        out = orig_stats(adv)
        seen["stats"] = (adv.detach().cpu().numpy().copy(), [float(x) for x in out.reshape(-1)[:2].cpu()])
        return out

    monkeypatch.setattr(ops, "gae_scan", spy_gae)
    monkeypatch.setattr(ops, "adv_stats", spy_stats)
    obss = env.reset()
    for _ in range(T):
        obss = bench.one_step(agent, env, obss, N)
    assert agent.n_updates > 0 and "gae" in seen and "stats" in seen
    reward, v_pred, next_v, nonterm, cut, adv, vt = seen["gae"]
    gamma, lambd, mode = seen["gae_args"]
    assert reward.shape == (T, N) and cut[-1].all()
    n_frag = 0
    for e in range(N):
        a = 0
        for t in range(T):
            if cut[t, e]:
                wa, wv = oracle.gae_fragment(reward[a:t + 1, e], v_pred[a:t + 1, e],
                                             next_v[a:t + 1, e], nonterm[a:t + 1, e], gamma, lambd, mode)
                assert np.array_equal(adv[a:t + 1, e], wa.astype(adv.dtype)), (e, a, t)
                assert np.array_equal(vt[a:t + 1, e], wv.astype(vt.dtype)), (e, a, t)
                a = t + 1
                n_frag += 1
    assert n_frag > N        # episode ends inside the rollout: more fragments than envs
    flat, (mean, std) = seen["stats"]
    wm, ws = oracle.adv_stats(flat)
    assert abs(mean - wm) <= 1e-6 * max(1.0, abs(wm)) and abs(std - ws) <= 1e-6 * max(1.0, abs(ws))


@pytest.mark.parametrize("N,T,chunk", [(512, 128, 16384), (64, 24, 4096), (64, 24, 512), (64, 20, 512)])
def test_ppo_next_state_values_evaluated_once_are_the_full_second_pass_bit_for_bit(monkeypatch, N, T,
                                                                                    chunk):
    """The reference evaluates the model on ALL states and ALL next_states of a rollout
    (pfrl/agents/ppo.py:119-133).  The device path (default ``reuse_next_values=False``) takes the
    value of a next_state that IS the next step's state from the first pass and evaluates only
    the other rows (episode ends, the rollout's last step), planned as a full chunk
    (``mfma_trunk.plan_batch``): V(next_state) must equal the brute-force second pass over every
    next_state BIT FOR BIT -- at BASELINE configs[3]'s size, in one chunk, in several chunks --
    and with ragged chunks (no guarantee) the full pass must be what runs."""
    import bench

    dev = torch.device("cuda:0")
    args = _bench_args(algo="ppo", num_envs=N)
    agent, env, _ = bench.build_agent(args, dev, 0)
    assert agent.reuse_next_values is False, "the default must be the reference's semantics"
    agent.update_interval = N * T
    agent.minibatch_size = N * T // 4
    agent.value_pass_chunk = chunk
    seen = {}
    orig = agent._next_values

    def spy(ro, T_, N_, v_pred, n_refs):
        got = orig(ro, T_, N_, v_pred, n_refs)
        seen["pass"] = dict(agent.next_value_pass)
        _, full = agent._value_pass(n_refs, None)
        seen["equal"] = bool(torch.equal(got, full))
        seen["max_abs"] = float((got - full).abs().max())
        seen["cuts"] = int(ro.h_cut[:T_ - 1].sum())
        return got

    monkeypatch.setattr(agent, "_next_values", spy)
    obss = env.reset()
    for _ in range(T):
        obss = bench.one_step(agent, env, obss, N)
    assert agent.n_updates > 0 and "pass" in seen
    assert seen["equal"], seen
    M = N * T
    if M <= chunk or M % chunk == 0:
        assert seen["pass"]["mode"].startswith("rows shared"), seen
        # one row per env for the last step + one per episode end, padded to a multiple of N
        assert seen["pass"]["evaluated"] <= N + seen["cuts"] + N, seen
    else:
        assert seen["pass"]["mode"] == "full pass" and seen["pass"]["evaluated"] == M, seen


def test_bench_per_launch_table_is_measured_and_leaves_the_agent_untouched():
    """``roofline.mfma.per_launch`` of the bench line (measured in the run, not
    read from a committed profile): the launches of ONE update -- the Python a capture records,
    run eagerly with an event pair around every library launch -- are the ten of the chain,
    carry 1.585 GFLOP at B = 32 in all, and parameters / optimizer state are what they were
    before the measurement."""
    import bench

    dev = torch.device("cuda:0")
    N = 256
    args = _bench_args(capacity=3000, frame_slots=12288, slack=512, replay_start=1024)
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    obss = env.reset()
    for _ in range(8):
        obss = bench.one_step(agent, env, obss, N)
    assert agent._graphed is not None and agent.optim_t > 0
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in agent.model.parameters()]
    state = [v.detach().clone() for st in agent.optimizer.state.values() for v in st.values()
             if torch.is_tensor(v)]
    table = bench.mfma_per_launch(agent, rbuf, 32)
    rows = table["launches"]
    assert table["n_launches"] == len(rows) == 10, [r["entry"] for r in rows]
    assert [r["entry"] for r in rows][:4] == ["pfrl_conv2d_nhwc_fwd"] * 4
    assert rows[4]["entry"] == "pfrl_dqn_head_td_loss" and rows[-1]["entry"] == "pfrl_rmsprop_fused_step"
    assert abs(sum(r["gflop"] for r in rows) - 1.585) < 0.01
    assert all(r["us"] > 0 and (r["frac"] is None or 0.0 < r["frac"] < 1.0) for r in rows)
    for a, b in zip(before, agent.model.parameters()):
        assert torch.equal(a, b)
    after = [v for st in agent.optimizer.state.values() for v in st.values() if torch.is_tensor(v)]
    for a, b in zip(state, after):
        assert torch.equal(a, b)
    # ... and training goes on
    obss = bench.one_step(agent, env, obss, N)


def test_dqn_act_graph_and_fused_head_equal_the_eager_act_path(monkeypatch):
    """``DQN.batch_act`` of the device step as one captured graph with the fused head
    (agents/_dqn_device_step.py::ActGraph, pfrl_dqn_act_head) against the launches it replaces
    (gather, trunk, narrow head, torch argmax, cast, pfrl_select_actions): on the bench agent,
    with updates running, the actions of every step and the parameters after the run are
    bit-identical, and the NumPy stream ends at the same position."""
    import bench

    dev = torch.device("cuda:0")
    N = 256

    def run(graph):
        monkeypatch.setenv("PFRL_DQN_ACT_GRAPH", "1" if graph else "0")
        monkeypatch.setenv("PFRL_DQN_ACT_HEAD", "1" if graph else "0")
        args = _bench_args(capacity=3000, frame_slots=12288, slack=512, replay_start=1024)
        agent, env, rbuf = bench.build_agent(args, dev, 0)
        obss = env.reset()
        acts = []
        for _ in range(24):
            a = agent.batch_act(obss)
            acts.append(np.asarray(a).copy())
            obss2, rs, dones, _ = env.step(a)
            agent.batch_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
            obss = env.reset(~np.asarray(dones))
        torch.cuda.synchronize()
        params = [p.detach().cpu().clone() for p in agent.model.parameters()]
        used = agent.__dict__.get("_act_graph")
        return acts, params, np.random.get_state(), used, agent.optim_t

    a0, p0, s0, g0, n0 = run(False)
    a1, p1, s1, g1, n1 = run(True)
    assert g0 is None and g1 is not None and g1.entries, "the graph path was not taken"
    assert n0 == n1 and n0 > 10 * 64
    for t, (x, y) in enumerate(zip(a0, a1)):
        np.testing.assert_array_equal(x, y, err_msg="step %d" % t)
    assert _same_rng_state(s0, s1)
    for x, y in zip(p0, p1):
        assert torch.equal(x, y)


def test_ppo_act_graph_equals_the_eager_act_path(monkeypatch):
    """The rollout's ``batch_act`` as one captured graph (agents/ppo.py::_ActGraph) against the
    eager launches it replaces, on the bench agent: entropy and value to f32 rounding, actions inside
    the action set, drawn anew on every replay (the generator advances) and distributed like the
    policy's probabilities; the rollout built from graph steps trains (an update runs)."""
    import bench
    from pfrl_amd.agents.ppo import _ActGraph
    from pfrl_amd.utils.contexts import evaluating

    dev = torch.device("cuda:0")
    N = 512
    args = _bench_args(algo="ppo", num_envs=N)
    agent, env, _ = bench.build_agent(args, dev, 0)
    obss = env.reset()
    for _ in range(3):
        obss = bench.one_step(agent, env, obss, N)
    assert agent._act_graph is not None and agent._act_graph.entries, "the graph path was not taken"
    refs_dev = torch.as_tensor(obss.refs, device=dev)
    with torch.no_grad(), evaluating(agent.model):
        distrib, value = agent.model(agent._features(refs_dev))
        want_entropy, want_value = distrib.entropy().float(), value.reshape(-1).float()
        probs = distrib.probs.double().mean(dim=0).cpu().numpy()
    counts = np.zeros(probs.shape[0])
    prev = None
    for i in range(40):
        action, stats = agent._act_graph.run(refs_dev)
        a = action.clone()
        # (one launch for heads + sampling + entropy: same numbers to f32 rounding of another
        # summation order, 512-term dot products by wave reduction)
        assert torch.allclose(stats[0], want_entropy, rtol=1e-5, atol=2e-6)
        assert torch.allclose(stats[1], want_value, rtol=1e-5, atol=2e-6)
        assert int(a.min()) >= 0 and int(a.max()) < probs.shape[0]
        if prev is not None:
            assert not torch.equal(a, prev)
        prev = a
        counts += np.bincount(a.cpu().numpy(), minlength=probs.shape[0])
    freq = counts / counts.sum()
    assert np.abs(freq - probs).max() < 0.02, (freq, probs)     # 20 480 draws
    # a subclass / instance that overrides the sampling hook keeps the eager path
    monkeypatch.setattr(agent, "_sample_action", lambda d: d.sample(), raising=False)
    assert not _ActGraph(agent).applicable()


def test_ppo_act_step_into_the_rollout_columns_equals_the_copying_step(monkeypatch):
    """Round 6: a captured acting step that writes its outputs where they belong -- actions into the
    rollout's action column, (entropy, value) into the ring the statistics windows read, frame slots
    and the two row indices arriving in ONE staging transfer (``_ActGraph.run_in_place``) --
    against the round-5 step (graph outputs cloned, the action copied into its column): same
    generator use, so two rollouts + updates give the same actions at every step, the same
    statistics and bit-identical parameters."""
    import bench

    dev = torch.device("cuda:0")
    N, T = 64, 24

    def run(in_place):
        monkeypatch.setenv("PFRL_PPO_ACT_IN_PLACE", "1" if in_place else "0")
        args = _bench_args(algo="ppo", num_envs=N)
        agent, env, _ = bench.build_agent(args, dev, 0)
        agent.update_interval = N * T
        agent.minibatch_size = N * T // 4
        torch.manual_seed(11)
        obss = env.reset()
        acts = []
        for step in range(2 * T + 3):
            if in_place and step == 7:
                # a capture in the MIDDLE of a rollout (a changed module tree would cause one): its
                # warm-up launches must not touch rows that already hold the rollout's data
                agent._act_graph.entries.clear()
            a = agent.batch_act(obss)
            acts.append(np.asarray(a).copy())
            obss, rs, dones, _ = env.step(a)
            agent.batch_observe(obss, rs, dones, np.zeros(N, dtype=bool))
            obss = env.reset(~np.asarray(dones))
        torch.cuda.synchronize()
        stats = dict(agent.get_statistics())
        return acts, [p.detach().cpu().clone() for p in agent.model.parameters()], stats, agent

    a0, p0, s0, ag0 = run(False)
    a1, p1, s1, ag1 = run(True)
    assert ag1._stats_ring_buf is not None and ag0._stats_ring_buf is None
    assert ag0.n_updates == ag1.n_updates == 2 * 4 * 4
    for t, (x, y) in enumerate(zip(a0, a1)):
        np.testing.assert_array_equal(x, y, err_msg="step %d" % t)
    for x, y in zip(p0, p1):
        assert torch.equal(x, y)
    for k in ("average_value", "average_entropy", "average_value_loss", "average_policy_loss"):
        assert s0[k] == s1[k], (k, s0[k], s1[k])


def test_ppo_captured_minibatch_update_equals_the_eager_update(monkeypatch):
    """One minibatch update of PPO (gather, forward, loss, backward, clipping, Adam) replayed from
    a HIP graph (agents/ppo.py::_minibatch_step, default) against the eager launch sequence
    (PFRL_PPO_UPDATE_GRAPH=0) on identical rollouts (the sampled actions are replayed from a
    table): same parameters and loss statistics after three rollouts of 2 epochs x 4 minibatches.
    Tolerance 2e-6: torch's capturable Adam keeps its step count on the device and forms the bias
    corrections there in f32."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.nn import Branched
    from pfrl_amd.policies import SoftmaxCategoricalHead

    dev = torch.device("cuda:0")
    N, T, rollouts = 32, 16, 3
    table = np.random.RandomState(7).randint(0, 6, size=(T * rollouts + 1, N))

    def run(graph):
        monkeypatch.setenv("PFRL_PPO_UPDATE_GRAPH", "1" if graph else "0")
        pfrl.utils.set_random_seed(0)
        torch.manual_seed(99)
        model = torch.nn.Sequential(
            torch.nn.Flatten(), torch.nn.Linear(4 * 144, 64), torch.nn.ReLU(),
            Branched(torch.nn.Sequential(torch.nn.Linear(64, 6), SoftmaxCategoricalHead()),
                     torch.nn.Linear(64, 1)))
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-5, fused=True)
        store = DeviceFrameStore((T + 8) * N + 512, (12, 12), torch.uint8, dev, stack=4)
        env = SyntheticAtariVectorEnv(N, store=store, seed=3, frame_shape=(12, 12), p_done=0.05)
        ag = agents.PPO(model, opt, gpu=0, phi=lambda x: np.asarray(x, dtype=np.float32) / 255,
                        update_interval=N * T, minibatch_size=N * T // 4, epochs=2, clip_eps=0.1,
                        standardize_advantages=True, entropy_coef=1e-2, max_grad_norm=0.5)
        step = [0]

        def replay_action(distrib):
            a = torch.as_tensor(table[step[0]], device=dev)
            step[0] += 1
            return a

        ag._sample_action = replay_action
        obss = env.reset()
        for _ in range(T * rollouts):
            a = ag.batch_act(obss)
            obss, r, d, _ = env.step(a)
            ag.batch_observe(obss, r, d, np.zeros(N, dtype=bool))
            obss = env.reset(~d)
        torch.cuda.synchronize()
        assert ag.n_updates == rollouts * 8
        assert (ag._update_graph is not None) == graph
        return (np.concatenate([p.detach().cpu().numpy().ravel() for p in model.parameters()]),
                ag.value_loss_record.values(), ag.policy_loss_record.values())

    pa, va, la = run(True)
    pb, vb, lb = run(False)
    np.testing.assert_allclose(pa, pb, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(va, vb, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(la, lb, rtol=1e-5, atol=1e-6)


def test_ppo_schedules_on_lr_and_clip_eps_reach_the_captured_update(monkeypatch):
    """the captured minibatch update used to bake ``clip_eps`` in and to
    re-capture every rollout under an lr schedule.  Now the graph is keyed by the Python-side
    numbers it bakes (a changed ``clip_eps`` captures anew) and reads the learning rate from a
    device scalar (a changed lr does NOT): a run with both schedules equals the eager run, with one
    graph per distinct clip_eps."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.nn import Branched
    from pfrl_amd.policies import SoftmaxCategoricalHead

    dev = torch.device("cuda:0")
    N, T, rollouts = 32, 16, 4
    table = np.random.RandomState(7).randint(0, 6, size=(T * rollouts + 1, N))

    def run(graph):
        monkeypatch.setenv("PFRL_PPO_UPDATE_GRAPH", "1" if graph else "0")
        pfrl.utils.set_random_seed(0)
        torch.manual_seed(99)
        model = torch.nn.Sequential(
            torch.nn.Flatten(), torch.nn.Linear(4 * 144, 64), torch.nn.ReLU(),
            Branched(torch.nn.Sequential(torch.nn.Linear(64, 6), SoftmaxCategoricalHead()),
                     torch.nn.Linear(64, 1)))
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-5, fused=True)
        store = DeviceFrameStore((T + 8) * N + 512, (12, 12), torch.uint8, dev, stack=4)
        env = SyntheticAtariVectorEnv(N, store=store, seed=3, frame_shape=(12, 12), p_done=0.05)
        ag = agents.PPO(model, opt, gpu=0, phi=lambda x: np.asarray(x, dtype=np.float32) / 255,
                        update_interval=N * T, minibatch_size=N * T // 4, epochs=2, clip_eps=0.2,
                        standardize_advantages=True, entropy_coef=1e-2, max_grad_norm=0.5)
        step = [0]

        def replay_action(distrib):
            a = torch.as_tensor(table[step[0]], device=dev)
            step[0] += 1
            return a

        ag._sample_action = replay_action
        obss = env.reset()
        for t in range(T * rollouts):
            # what LinearInterpolationHook does before every step (train_ppo_ale.py:301-311)
            for g in opt.param_groups:
                g["lr"] = 1e-3 * (1.0 - 0.5 * t / (T * rollouts))
            ag.clip_eps = 0.2 if t < 2 * T else 0.1
            a = ag.batch_act(obss)
            obss, r, d, _ = env.step(a)
            ag.batch_observe(obss, r, d, np.zeros(N, dtype=bool))
            obss = env.reset(~d)
        torch.cuda.synchronize()
        assert ag.n_updates == rollouts * 8
        n_graphs = len(ag._update_graph.graphs) if graph else 0
        assert all(isinstance(g["lr"], float) for g in opt.param_groups)
        return (np.concatenate([p.detach().cpu().numpy().ravel() for p in model.parameters()]),
                ag.value_loss_record.values(), ag.policy_loss_record.values(), n_graphs)

    pa, va, la, n_graphs = run(True)
    pb, vb, lb, _ = run(False)
    assert n_graphs == 2            # one per clip_eps value; four different learning rates
    np.testing.assert_allclose(pa, pb, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(va, vb, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(la, lb, rtol=1e-5, atol=1e-6)


def _rainbow_run(monkeypatch, feed, steps=10):
    import bench

    monkeypatch.setenv("PFRL_NOISE_FEED", "1" if feed else "0")
    dev = torch.device("cuda:0")
    N, CAP, START = 256, 3000, 1024
    args = _bench_args(algo="rainbow", capacity=CAP, frame_slots=3000 + 24 * N + 512, slack=1024,
                       replay_start=START)
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    obss = env.reset()
    actions = []
    for _ in range(steps):
        a = agent.batch_act(obss)
        actions.append(torch.as_tensor(a.tensor if hasattr(a, "tensor") else np.asarray(a)).cpu().numpy().copy())
        obss, rs, dones, infos = env.step(a)
        agent.batch_observe(obss, rs, dones, np.zeros(N, dtype=bool))
        obss = env.reset(np.logical_not(dones))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in agent.model.parameters()]).cpu()
    tparams = torch.cat([p.detach().reshape(-1) for p in agent.target_model.parameters()]).cpu()
    fed = [e.get("noise") for e in agent._graphed.graphs.values() if isinstance(e, dict)]
    return dict(actions=np.asarray(actions), params=params, tparams=tparams, optim_t=agent.optim_t,
                rng=torch.cuda.get_rng_state(dev).clone(), fed=fed)


def test_rainbow_noise_draws_as_one_launch_equal_the_torch_randn_calls(monkeypatch):
    """the nine factorised-noise draws of a Rainbow update (three NoisyNet
    layers x three passes; the reference draws each with torch.normal on the device generator in
    call order, pfrl/nn/noisy_linear.py:52-60) come from ONE launch in front of the update's graph
    (csrc/philox.hip, graphed_update._fill_noise).  Gate: the run is the run with the torch.randn
    calls inside the graph -- every acting draw in between sees the same generator state, so
    actions, online and target parameters and the final generator state are BIT-IDENTICAL."""
    a = _rainbow_run(monkeypatch, feed=False)
    b = _rainbow_run(monkeypatch, feed=True)
    assert a["optim_t"] == b["optim_t"] >= 100
    assert not any(a["fed"]) and all(n is not None and len(n["sizes"]) == 9 for n in b["fed"]) and b["fed"]
    assert b["fed"][0]["sizes"] == [3136 + 1024, 512 + 6 * 51, 512 + 51] * 3
    assert torch.equal(a["rng"], b["rng"])
    np.testing.assert_array_equal(a["actions"], b["actions"])
    assert torch.equal(a["params"], b["params"]) and torch.equal(a["tparams"], b["tparams"])


@pytest.mark.gpu
def test_ppo_update_on_u8_minibatches_equals_the_fp32_minibatches(monkeypatch):
    """configs[3]'s model and update as bench.py builds them (64 envs x 128 steps: minibatches of
    2 048 observations): the captured update with the first convolution reading u8 NHWC4
    minibatches (agents/ppo.py ``_u8_pixels``: acting, value pass and update; default) leaves the SAME parameters, bit for
    bit, as with the fp32 minibatch gathered first (PFRL_U8_CONV1=0)."""
    import bench
    from pfrl_amd.nn import mfma_trunk
    from pfrl_amd import ops

    dev = torch.device("cuda:0")
    N, T = 64, 128

    def run(u8):
        monkeypatch.setattr(mfma_trunk, "_U8_FIRST", u8)
        args = _bench_args(algo="ppo", num_envs=N)
        agent, env, _ = bench.build_agent(args, dev, 0)
        seen = []
        orig = ops.batch_states_raw_nhwc4

        def spy(*a, **kw):
            seen.append(1)
            return orig(*a, **kw)

        monkeypatch.setattr(ops, "batch_states_raw_nhwc4", spy)
        obss = env.reset()
        for _ in range(2 * T):
            obss = bench.one_step(agent, env, obss, N)
        torch.cuda.synchronize()
        monkeypatch.setattr(ops, "batch_states_raw_nhwc4", orig)
        assert agent.n_updates == 2 * 4 * 4       # two rollouts x 4 epochs x 4 minibatches
        assert bool(seen) == u8
        return np.concatenate([p.detach().cpu().numpy().ravel() for p in agent.model.parameters()])

    a = run(True)
    b = run(False)
    assert np.array_equal(a, b)
