"""Parity at BASELINE.json's EXACT sizes: configs[1] / configs[2] with a
10^6-transition replay -- a 7.1 GB frame ring whose byte offsets exceed 2^32, entry / transition
rings of 1 065 536 rows, priority trees whose index frame is 2^20 wide and re-roots to 2^21 --
and configs[4] at its bench shape (obs f32[376], action f32[17], B = 256, 64 envs, the 2 048-
and 14 336-entry launches).  Built exactly as ``bench.py`` builds them, filled through the
normal act / observe path, then compared with an expectation made independently of the
product: the reference's loop restated on plain arrays (pfrl/agents/dqn.py:516-549,
pfrl/agents/soft_actor_critic.py:354-374), index sets by a restatement of
pfrl/utils/random.py:4-28 on the same NumPy stream position, minibatches by the C oracle.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _ref_sample_n_k(n, k):
    """pfrl/utils/random.py:4-28."""
    if 3 * k >= n:
        return np.random.choice(n, k, replace=False)
    result = np.random.choice(n, 2 * k)
    selected = set()
    j = k
    for i in range(k):
        x = result[i]
        while x in selected:
            x = result[i] = result[j]
            j += 1
            if j == 2 * k:
                result[k:] = np.random.choice(n, k)
                j = k
        selected.add(x)
    return result[:k]


def _same_rng_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_dqn_configs1_at_capacity_1e6_matches_oracle():
    """configs[1], C = 10^6: every transition of the run is recorded on the host as plain
    columns (what pfrl's ReplayBuffer(10**6).memory would hold); after the buffer is full and
    its rings have wrapped, three checked steps compare all 64 minibatches (index sets through
    the gathered bytes, action / reward / terminal / discount, fp32 stacks) with the oracle bit
    for bit.  Sampled frames lie above byte offset 2^32 of the ring."""
    import bench
    import oracle
    from test_bench_path_parity import _bench_args

    dev = torch.device("cuda:0")
    N, B, CAP = 256, 32, 10 ** 6
    args = _bench_args()                       # bench.py's defaults: capacity 10**6
    assert args.capacity == CAP
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    store = env.store
    fb = store.frame_bytes
    assert store.n_slots * fb > 7 * 10 ** 9
    n_steps = CAP // N + 260                   # fill, then 260 more steps: E = R = C + 65536 wrap
    S = np.zeros((n_steps + 8, N, 4), np.int32)
    NX = np.zeros((n_steps + 8, N, 4), np.int32)
    A = np.zeros((n_steps + 8, N), np.int64)
    R = np.zeros((n_steps + 8, N), np.float64)
    D = np.zeros((n_steps + 8, N), np.uint8)

    fetched, keep = [], [False]
    orig_slots = None

    def install_spy():
        nonlocal orig_slots
        orig_slots = rbuf.store.fetch_many_slots

        def spy(slots_dev, U, B_, phi, g):
            big = orig_slots(slots_dev, U, B_, phi, g)
            if keep[0]:
                fetched.append(({k: v.detach().cpu().numpy() for k, v in big.items()},
                                slots_dev.cpu().numpy().reshape(U, B_).copy()))
            return big

        rbuf.store.fetch_many_slots = spy

    obss = env.reset()
    # updates off while filling (as bench.prefill does), on for the last 4 steps
    saved_start = agent.replay_updater.replay_start_size
    agent.replay_updater.replay_start_size = 1 << 62
    checked = 0
    max_byte_offset = 0
    for step in range(n_steps + 4):
        last = step >= n_steps
        if step == n_steps:
            agent.replay_updater.replay_start_size = saved_start
            install_spy()
        S[step] = obss.refs
        acts = agent.batch_act(obss)
        obss2, rs, dones, _ = env.step(acts)
        NX[step] = obss2.refs
        if last or step % 64 == 0:
            A[step] = np.asarray(acts)      # (D2H; the fill steps keep their actions on the device)
        R[step], D[step] = rs, dones
        keep[0] = last
        del fetched[:]
        s0 = np.random.get_state()
        agent.batch_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
        s1 = np.random.get_state()
        if last:
            torch.cuda.synchronize()
            # expectation: the reference's loop on the recorded columns
            np.random.set_state(s0)
            total0 = step * N                         # transitions appended before this step
            expected = []
            for i in range(N):
                total = total0 + i + 1
                ln = min(total, CAP)
                if ln >= saved_start and (total % 4) == 0:
                    idx = _ref_sample_n_k(ln, B)
                    expected.append(total - ln + np.asarray(idx, dtype=np.int64))   # global ids
            assert _same_rng_state(np.random.get_state(), s1), "NumPy stream position differs"
            assert len(expected) == 64 and len(rbuf) == CAP
            got = {k: np.concatenate([f[0][k] for f in fetched]) for k in fetched[0][0]
                   if k != "target_next_raw"}
            slots = np.concatenate([f[1] for f in fetched])
            assert got["state"].shape[:2] == (64, B)
            # the frames the expectation needs, fetched by slot (the ring itself is 7.1 GB)
            for u, gids in enumerate(expected):
                st_, en_ = gids // N, gids % N
                s_refs, n_refs = S[st_, en_], NX[st_, en_]
                need = np.unique(np.concatenate([s_refs.ravel(), n_refs.ravel()]))
                max_byte_offset = max(max_byte_offset, int(need.max()) * fb)
                fr = store.frames[torch.from_numpy(need.astype(np.int64)).to(dev)].cpu().numpy()
                fr = fr.reshape(len(need), -1)
                remap = {int(s): j for j, s in enumerate(need)}
                loc = np.vectorize(remap.get)
                want_s = oracle.batch_states_u8(fr, loc(s_refs).astype(np.int32), 255.0)
                want_n = oracle.batch_states_u8(fr, loc(n_refs).astype(np.int32), 255.0)
                # channels_last minibatch buffers hold the same values: compare as NCHW
                assert np.array_equal(np.ascontiguousarray(got["state"][u]).reshape(B, -1),
                                      want_s.reshape(B, -1)), (step, u)
                assert np.array_equal(np.ascontiguousarray(got["next_state"][u]).reshape(B, -1),
                                      want_n.reshape(B, -1)), (step, u)
                # entry ring slot of global id g is g % E (entries are one transition long)
                assert np.array_equal(slots[u], (gids % rbuf.store.E).astype(np.int32))
                sc = oracle.batch_experiences_scalars([[b] for b in range(B)], R[st_, en_],
                                                      D[st_, en_], agent.gamma, 1)
                for key in ("reward", "is_state_terminal", "discount"):
                    assert np.array_equal(got[key][u], sc[key]), (step, u, key)
                known = (st_ >= n_steps) | (st_ % 64 == 0)     # steps whose actions were read
                assert np.array_equal(got["action"][u][known], A[st_, en_][known])
            checked += 1
        obss = env.reset(~np.asarray(dones))
    assert checked == 4
    assert max_byte_offset > 2 ** 32, "no sampled frame above 4 GiB of the ring"
    st = rbuf.store
    assert st.E == st.R == CAP + 65536 and st.n_entries > st.E      # entry / transition rings wrapped
    assert store.next_seq > store.n_slots                            # and the frame ring


def test_rainbow_configs2_at_capacity_1e6_tree_matches_oracle():
    """configs[2], C = 10^6, in the mode ``bench.py --algo rainbow`` runs: the priority trees'
    index frame grows to 2^20 leaves while filling, slides, and re-roots to 2^21 and back once
    the buffer is full; for the checked steps every sample's indices and priorities and the
    tree root (sum / min / max_priority with type tags) after every update_errors equal
    ``OraclePrioritizedBuffer`` fed the same appends, draws and TD errors."""
    import bench
    import oracle
    from pfrl_amd.collections import prioritized as dev_pri
    from test_bench_path_parity import _bench_args

    dev = torch.device("cuda:0")
    N, CAP = 256, 10 ** 6
    args = _bench_args(algo="rainbow")
    assert args.capacity == CAP and args.priority_pow == "device"
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    orc = oracle.OraclePrioritizedBuffer(CAP)
    counts = dict(samples=0, updates=0)
    log2_seen = set()
    checking = [False]

    orig_append = dev_pri.PrioritizedBuffer.append
    orig_sample = dev_pri.PrioritizedBuffer.sample_device

    # (the host runs one update point ahead of the device, DQN._batch_observe_train_per: the oracle
    # gets the calls in the reference's logical order -- see test_rainbow_bench_path_tree_matches_oracle)
    late_appends = []

    def spy_append(self, value, priority=None):
        if self.flag_wait_priority:
            late_appends.append(value)
        else:
            orc.append(value)
        r = orig_append(self, value, priority)
        log2_seen.add(self.frame.log2_size)
        return r

    def check_sample(self, out, want, beta):
        if checking[0]:
            self._join()
            x = out["x"].cpu().numpy()
            np.testing.assert_array_equal(x - self.frame.head, want["indices"])
            np.testing.assert_array_equal(out["pri"].cpu().numpy(), want["priorities"])
            w = (want["probabilities"] / want["min_prob"]) ** (-beta)
            np.testing.assert_allclose(out["weight"].cpu().numpy(), w, rtol=1e-5)
            counts["samples"] += 1

    def spy_sample(self, n, u01=None, normalize=1, beta=0.0, slot_mod=0, **kw):
        u = np.random.random_sample(n)
        if kw.get("split"):
            out, finish = orig_sample(self, n, u01=u, normalize=normalize, beta=beta,
                                      slot_mod=slot_mod, **kw)

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
        if checking[0]:
            if not late_appends:
                got, so = rbuf.memory.tree.root_stats(), orc.stats()
                assert got[0] == so["sum"] and got[1] == so["min"] and got[2] == so["max_priority"], \
                    (counts, got, so)
            counts["updates"] += 1
        for value in late_appends:
            orc.append(value)
        del late_appends[:]

    rbuf.update_errors = spy_update
    try:
        obss = env.reset()
        saved = agent.replay_updater.replay_start_size
        agent.replay_updater.replay_start_size = 1 << 62
        # fill to capacity (n = 3 windows: one entry per transition in steady state), then
        # enough further steps for the frame to slide past a power of two and re-root
        steps_fill = CAP // N + 8
        for step in range(steps_fill + 200):
            if step == steps_fill + 197:
                agent.replay_updater.replay_start_size = saved
                checking[0] = True
            obss = bench.one_step(agent, env, obss, N)
    finally:
        dev_pri.PrioritizedBuffer.append = orig_append
        dev_pri.PrioritizedBuffer.sample_device = orig_sample
    assert counts["updates"] >= 3 * 64 - 2 and counts["samples"] >= counts["updates"]
    assert len(rbuf) == len(orc) == CAP
    assert {20, 21} <= log2_seen, log2_seen            # the frame reached 2^20 and re-rooted to 2^21
    tree = rbuf.memory.tree
    assert tree.frame.log2_size in (20, 21)


def _spy_planned_gather(rbuf, fetched, keep):
    """The same record as the ``fetch_many`` spies below for ranges planned natively
    (agents/_vector_device_step.py): their gather goes through ``store.fetch_many_slots``."""
    st = rbuf.store
    orig = st.fetch_many_slots

    def spy(slots_dev, U, B, phi, g, **kw):
        big = orig(slots_dev, U, B, phi, g, **kw)
        if keep[0]:
            fetched.append(({k: v.detach().cpu().numpy() for k, v in big.items()
                             if isinstance(v, torch.Tensor)}, U))
        return big

    st.fetch_many_slots = spy


def test_sac_configs4_bench_shape_matches_oracle():
    """configs[4] as bench.py builds it (obs f32[376], action f32[17], B = 256, 64 host envs,
    update_interval 1, two env ranges -> 2 048- and 14 336-entry gathers): appends and index
    draws in the reference's order (pfrl/agents/soft_actor_critic.py:354-374); for checked steps
    the 64 minibatches -- observation rows, actions, reward / terminal / discount -- equal the
    recorded transitions bit for bit and the NumPy stream ends where the reference's loop ends."""
    import bench
    from test_bench_path_parity import _bench_args

    dev = torch.device("cuda:0")
    N, B, CAP = 64, 256, 20000
    args = _bench_args(algo="sac", num_envs=N, minibatch=B, capacity=CAP, blas="default")
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    assert tuple(agent.step_fused_chunks) == (0.125,)
    cols = dict(s=[], a=[], r=[], ns=[], d=[])
    fetched, keep = [], [False]
    orig_fetch = rbuf.fetch_many

    def spy_fetch(seq_sets, phi, g):
        big = orig_fetch(seq_sets, phi, g)
        if keep[0]:
            fetched.append(({k: v.detach().cpu().numpy() for k, v in big.items()
                             if isinstance(v, torch.Tensor)}, len(seq_sets)))
        return big

    rbuf.fetch_many = spy_fetch
    _spy_planned_gather(rbuf, fetched, keep)
    obss = env.reset()
    start = agent.replay_updater.replay_start_size
    n_steps = start // N + 6
    checked = 0
    shapes = set()
    for step in range(n_steps):
        acts = agent.batch_act(obss)
        obss2, rs, dones, _ = env.step(acts)
        t_before = len(cols["r"])
        for i in range(N):
            cols["s"].append(np.asarray(obss[i], dtype=np.float32))
            cols["a"].append(np.asarray(acts[i], dtype=np.float32))
            cols["r"].append(float(rs[i]))
            cols["ns"].append(np.asarray(obss2[i], dtype=np.float32))
            cols["d"].append(bool(dones[i]))
        keep[0] = True
        del fetched[:]
        s0 = np.random.get_state()
        agent.batch_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
        s1 = np.random.get_state()
        torch.cuda.synchronize()
        np.random.set_state(s0)
        expected = []
        for i in range(N):
            total = t_before + i + 1
            ln = min(total, CAP)
            if ln >= start:
                idx = _ref_sample_n_k(ln, B)
                expected.append(total - ln + np.asarray(idx, dtype=np.int64))
        # (the policy's sampling noise comes from torch's generator, not NumPy's)
        assert _same_rng_state(np.random.get_state(), s1), "NumPy stream position differs"
        if expected:
            assert sum(f[1] for f in fetched) == len(expected)
            if len(expected) == N:       # (the first updating step starts in mid-batch)
                assert [f[1] for f in fetched] == [8, 56]
                shapes |= {f[1] * B for f in fetched}
            got = {k: np.concatenate([f[0][k] for f in fetched]) for k in fetched[0][0]}
            S_, A_, NS_ = (np.stack(cols[k]) for k in ("s", "a", "ns"))
            R_, D_ = np.asarray(cols["r"]), np.asarray(cols["d"])
            for u, gids in enumerate(expected):
                assert np.array_equal(got["state"][u], S_[gids]), (step, u)
                assert np.array_equal(got["next_state"][u], NS_[gids]), (step, u)
                assert np.array_equal(got["action"][u], A_[gids]), (step, u)
                assert np.array_equal(got["reward"][u], R_[gids].astype(np.float32))
                assert np.array_equal(got["is_state_terminal"][u], D_[gids].astype(np.float32))
                assert np.array_equal(got["discount"][u],
                                      np.full(B, agent.gamma, dtype=np.float32))
            checked += 1
        obss = env.reset(~np.asarray(dones))
    assert checked >= 5 and shapes == {2048, 14336}


def test_sac_configs4_at_capacity_1e6_matches_oracle():
    """configs[4] at its STATED size: ReplayBuffer(10**6) of fp32 rows (obs f32[376], action f32[17]),
    filled to capacity through the agent's own act / observe path (64 host envs), then updating
    steps with the buffer wrapping: every sampled minibatch row -- checked through per-transition
    signatures (a fixed random projection of the observation rows, the action row, reward, terminal)
    kept for all 10**6 + transitions -- is the transition the reference's index draw
    (pfrl/replay_buffer.py:329-356 -> sample_n_k) names, and the NumPy stream ends where the
    reference's loop ends."""
    import bench
    from test_bench_path_parity import _bench_args

    dev = torch.device("cuda:0")
    N, B, CAP = 64, 256, 10 ** 6
    args = _bench_args(algo="sac", num_envs=N, minibatch=B, capacity=CAP, blas="default")
    agent, env, rbuf = bench.build_agent(args, dev, 0)
    proj_s = np.random.RandomState(1).randn(376)
    proj_a = np.random.RandomState(2).randn(17)
    sig = dict(s=[], a=[], r=[], ns=[], d=[])
    orig_observe = agent.batch_observe
    last = {}

    def note(obss2, rs, dones):
        sig["s"].append((np.stack([np.asarray(o, dtype=np.float32) for o in last["obs"]]).astype(np.float64) * proj_s).sum(axis=1))
        sig["a"].append((np.stack([np.asarray(a, dtype=np.float32) for a in last["act"]]).astype(np.float64) * proj_a).sum(axis=1))
        sig["ns"].append((np.stack([np.asarray(o, dtype=np.float32) for o in obss2]).astype(np.float64) * proj_s).sum(axis=1))
        sig["r"].append(np.asarray(rs, dtype=np.float64))
        sig["d"].append(np.asarray(dones, dtype=bool))

    # fill to capacity with updates off (bench.prefill's own switch), recording signatures
    saved = agent.replay_updater.replay_start_size
    agent.replay_updater.replay_start_size = 1 << 62
    obss = env.reset()
    while len(rbuf) < CAP:
        acts = agent.batch_act(obss)
        last["obs"], last["act"] = obss, acts
        obss2, rs, dones, _ = env.step(acts)
        note(obss2, rs, dones)
        orig_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
        obss = env.reset(~np.asarray(dones))
    agent.replay_updater.replay_start_size = saved
    assert len(rbuf) == CAP
    fetched = []
    orig_fetch = rbuf.fetch_many

    def spy_fetch(seq_sets, phi, g):
        big = orig_fetch(seq_sets, phi, g)
        fetched.append(({k: v.detach().cpu().numpy() for k, v in big.items()
                         if isinstance(v, torch.Tensor)}, len(seq_sets)))
        return big

    rbuf.fetch_many = spy_fetch
    _spy_planned_gather(rbuf, fetched, [True])
    checked = 0
    for step in range(4):
        acts = agent.batch_act(obss)
        last["obs"], last["act"] = obss, acts
        obss2, rs, dones, _ = env.step(acts)
        t_before = sum(len(x) for x in sig["r"])
        note(obss2, rs, dones)
        del fetched[:]
        s0 = np.random.get_state()
        orig_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
        s1 = np.random.get_state()
        torch.cuda.synchronize()
        np.random.set_state(s0)
        expected = []
        for i in range(N):
            total = t_before + i + 1
            ln = min(total, CAP)
            idx = _ref_sample_n_k(ln, B)
            expected.append(total - ln + np.asarray(idx, dtype=np.int64))
        assert _same_rng_state(np.random.get_state(), s1), "NumPy stream position differs"
        assert sum(f[1] for f in fetched) == N and [f[1] for f in fetched] == [8, 56]
        got = {k: np.concatenate([f[0][k] for f in fetched]) for k in fetched[0][0]}
        S_, A_, NS_ = (np.concatenate(sig[k]) for k in ("s", "a", "ns"))
        R_, D_ = np.concatenate(sig["r"]), np.concatenate(sig["d"])
        for u, gids in enumerate(expected):
            assert np.array_equal((got["state"][u].astype(np.float64) * proj_s).sum(axis=1), S_[gids]), (step, u)
            assert np.array_equal((got["next_state"][u].astype(np.float64) * proj_s).sum(axis=1), NS_[gids]), (step, u)
            assert np.array_equal((got["action"][u].astype(np.float64) * proj_a).sum(axis=1), A_[gids]), (step, u)
            assert np.array_equal(got["reward"][u], R_[gids].astype(np.float32))
            assert np.array_equal(got["is_state_terminal"][u], D_[gids].astype(np.float32))
        checked += 1
        obss = env.reset(~np.asarray(dones))
    assert checked == 4 and sum(len(x) for x in sig["r"]) > CAP      # the ring wrapped
