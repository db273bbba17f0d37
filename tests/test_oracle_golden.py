"""Pin the CPU oracle (oracle/pfrl_oracle.c) against vectors recorded from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import oracle
from oracle import OracleNStep, OraclePrioritizedBuffer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _files(prefix):
    fs = sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))
    assert fs, "no golden fixtures for " + prefix
    return fs


def _check_dump(buf, which, v, t):
    st = buf.stats()
    size = st["bounds"][1] - st["bounds"][0]
    off = 0
    w = 1
    while w <= size:
        n = size // w
        ov, ot = buf.dump_level(which, w)
        np.testing.assert_array_equal(ot, t[off:off + n])
        np.testing.assert_array_equal(ov, v[off:off + n])
        off += n
        w *= 2
    assert off == len(v)


@pytest.mark.parametrize("path", _files("pbuf_trace_"), ids=os.path.basename)
def test_prioritized_buffer_trace(path):
    """collections/prioritized.py: indices, removed priorities, tree sums/mins,
    max_priority and frame bounds are bit-exact after every operation."""
    g = np.load(path)
    cap = int(g["meta"][1])
    buf = OraclePrioritizedBuffer(None if cap < 0 else cap)
    iu = ia = iset = ismp = 0
    payload = 0
    for k, (kind, n) in enumerate(zip(g["op_kind"], g["op_n"])):
        if kind in (0, 1):
            v, t = g["app_v"][ia], int(g["app_t"][ia])
            ia += 1
            if t == 0:
                buf.append(payload)
            else:
                oracle.lib().orc_pbuf_append(buf._h, payload, float(v), t)
            payload += 1
        elif kind == 4:
            buf.popleft()
        else:
            r = buf.sample(g["u01"][iu:iu + n])
            np.testing.assert_array_equal(r["indices"], g["idx"][iu:iu + n])
            np.testing.assert_array_equal(r["priorities"], g["pri_v"][iu:iu + n])
            np.testing.assert_array_equal(r["priority_tags"], g["pri_t"][iu:iu + n])
            np.testing.assert_allclose(r["probabilities"], g["prob"][iu:iu + n], rtol=1e-6)
            assert r["total"] == g["total_v"][ismp] and r["total_tag"] == g["total_t"][ismp]
            np.testing.assert_allclose(r["min_prob"], g["min_prob"][ismp], rtol=1e-6)
            buf.set_last_priority(g["set_v"][iset:iset + n], g["set_t"][iset:iset + n])
            iu += n
            iset += n
            ismp += 1
        st = buf.stats()
        assert st["length"] == g["length"][k]
        if st["length"]:
            assert st["sum"] == (g["sum_v"][k], g["sum_t"][k]), k
            assert st["min"] == (g["min_v"][k], g["min_t"][k]), k
            assert st["bounds"] == (g["ixl"][k], g["ixr"][k]), k
        assert st["max_priority"] == (g["maxp_v"][k], g["maxp_t"][k]), k
    _check_dump(buf, 0, g["final_sum_v"], g["final_sum_t"])
    _check_dump(buf, 1, g["final_min_v"], g["final_min_t"])


@pytest.mark.parametrize("path", _files("per_trace_"), ids=os.path.basename)
def test_prioritized_replay_buffer_trace(path):
    """replay_buffers/prioritized.py on top of the n-step windows: sampled
    entries, priority transform (this host's libm, as the reference), weights."""
    g = np.load(path)
    seed, cap, n_steps, batch, n_envs = (int(x) for x in g["meta"])
    alpha, beta0, betasteps, eps = (float(x) for x in g["hyper"])
    norm = int(g["normalize_by_max"])
    buf = OraclePrioritizedBuffer(None if cap < 0 else cap)
    ns = OracleNStep(n_steps, max_envs=n_envs)
    entries = {}  # payload id -> list of tids
    next_payload = 0
    tid = 0
    iu = ie = ismp = idump = 0
    beta = beta0
    beta_add = (1.0 - beta0) / betasteps
    for k, (kind, a, b) in enumerate(zip(g["op_kind"], g["op_a"], g["op_b"])):
        if kind == 0:
            emitted = ns.append(a, tid, b)
            tid += 1
        elif kind == 1:
            emitted = ns.stop(a)
        else:
            emitted = []
            r = buf.sample(g["u01"][iu:iu + batch])
            np.testing.assert_array_equal(r["indices"], g["idx"][iu:iu + batch])
            assert (r["total"], r["total_tag"]) == (g["smp_total_v"][ismp], g["smp_total_t"][ismp])
            np.testing.assert_allclose(r["probabilities"], g["prob"][iu:iu + batch], rtol=1e-6)
            for j in range(batch):
                want = g["entry_tids"][(ie + j) * n_steps:(ie + j + 1) * n_steps]
                want = [int(x) for x in want if x >= 0]
                assert entries[int(r["payload"][j])] == want
            # weights (prioritized.py:57-66); float tolerance
            probs = r["probabilities"]
            if norm == 1:
                w = (probs / probs.min()) ** -beta
            elif norm == 2:
                w = (probs / r["min_prob"]) ** -beta
            else:
                w = (len(buf) * probs) ** -beta
            assert beta == pytest.approx(float(g["beta"][ismp]), abs=1e-12)
            beta = min(1.0, beta + beta_add)
            np.testing.assert_allclose(w, g["weight"][iu:iu + batch], rtol=2e-6)
            # priority transform
            err = g["err"][iu:iu + batch]
            is_py = g["err_is_py"][iu:iu + batch]
            pv = np.zeros(batch)
            pt = np.zeros(batch, dtype=np.int32)
            v32, t32 = oracle.priority_from_errors_f32(err.astype(np.float32), 0, 1, eps, alpha)
            for j in range(batch):
                if is_py[j]:
                    pv[j] = (min(1, max(0, float(err[j]))) + eps) ** alpha
                    pt[j] = 1
                else:
                    pv[j], pt[j] = v32[j], t32[j]
            np.testing.assert_array_equal(pt, g["new_pri_t"][iu:iu + batch])
            np.testing.assert_array_equal(pv, g["new_pri_v"][iu:iu + batch])
            buf.set_last_priority(pv, pt)
            iu += batch
            ie += batch
            ismp += 1
        for e in emitted:
            entries[next_payload] = [int(x) for x in e]
            buf.append(next_payload)
            next_payload += 1
        st = buf.stats()
        assert st["length"] == g["length"][k], k
        if st["length"]:
            assert st["sum"] == (g["sum_v"][k], g["sum_t"][k]), k
            assert st["min"] == (g["min_v"][k], g["min_t"][k]), k
            assert st["bounds"] == (g["ixl"][k], g["ixr"][k]), k
        assert st["max_priority"] == (g["maxp_v"][k], g["maxp_t"][k]), k
        if idump < len(g["dump_op"]) and g["dump_op"][idump] == k:
            lo, hi = g["dump_off"][idump], g["dump_off"][idump + 1]
            if hi > lo:
                _check_dump(buf, 0, g["dump_sum_v"][lo:hi], g["dump_sum_t"][lo:hi])
                _check_dump(buf, 1, g["dump_min_v"][lo:hi], g["dump_min_t"][lo:hi])
            idump += 1


@pytest.mark.parametrize("path", _files("replay_trace_"), ids=os.path.basename)
def test_uniform_replay_trace(path):
    """replay_buffers/replay_buffer.py n-step windows + FIFO capacity, and
    replay_buffer.py:157-212 batch_experiences scalars."""
    g = np.load(path)
    seed, cap, n_steps, n_envs, batch = (int(x) for x in g["meta"])
    gamma = float(g["gamma"])
    ns = OracleNStep(n_steps, max_envs=n_envs)
    fifo = []
    rewards, terminals = [], []
    tid = 0
    isample = 0
    ie = 0
    for k, (kind, a, b) in enumerate(zip(g["op_kind"], g["op_a"], g["op_b"])):
        if kind == 0:
            rewards.append(float(g["reward"][tid]))
            terminals.append(int(b))
            emitted = ns.append(a, tid, b)
            tid += 1
        else:
            emitted = ns.stop(a)
        for e in emitted:
            fifo.append([int(x) for x in e])
            if cap >= 0 and len(fifo) > cap:
                fifo.pop(0)
        assert len(fifo) == g["length"][k]
        if isample < len(g["s_at_op"]) and g["s_at_op"][isample] == k:
            idx = g["s_indices"][isample * batch:(isample + 1) * batch]
            ents = [fifo[int(i)] for i in idx]
            for j, e in enumerate(ents):
                want = g["s_entry_tids"][(ie + j) * n_steps:(ie + j + 1) * n_steps]
                assert e == [int(x) for x in want if x >= 0]
            r = oracle.batch_experiences_scalars(ents, rewards, terminals, gamma, n_steps)
            sl = slice(isample * batch, (isample + 1) * batch)
            np.testing.assert_array_equal(r["reward"], g["s_reward"][sl])
            np.testing.assert_array_equal(r["is_state_terminal"], g["s_terminal"][sl])
            np.testing.assert_array_equal(r["discount"], g["s_discount"][sl])
            np.testing.assert_array_equal(r["first"], g["s_state_tid"][sl])
            np.testing.assert_array_equal(r["last"] + 1, g["s_next_state_tid"][sl])
            isample += 1
            ie += batch
    assert [len(e) for e in fifo] == list(g["final_len"])
    for e, want in zip(fifo, g["final_tids"]):
        assert e == [int(x) for x in want if x >= 0]


def test_batch_states_atari_phi():
    g = np.load(os.path.join(GOLDEN, "batch_states_atari.npz"))
    frames = g["frames"]
    out = oracle.batch_states_u8(frames.reshape(len(frames), -1), g["refs"], 255.0)
    np.testing.assert_array_equal(out.reshape(g["out"].shape), g["out"])
    lut = oracle.batch_states_u8(np.arange(256, dtype=np.uint8).reshape(256, 1),
                                 np.arange(256, dtype=np.int32).reshape(256, 1), 255.0)
    np.testing.assert_array_equal(lut.ravel(), g["lut"])


def test_gae_fragments():
    g = np.load(os.path.join(GOLDEN, "gae.npz"))
    for c in range(len(g["mode"])):
        lo, hi = g["off"][c], g["off"][c + 1]
        adv, vt = oracle.gae_fragment(g["reward"][lo:hi], g["v"][lo:hi], g["nv"][lo:hi],
                                      g["nonterm"][lo:hi], g["gamma"][c], g["lambd"][c],
                                      int(g["mode"][c]))
        np.testing.assert_array_equal(adv, g["adv"][lo:hi])
        np.testing.assert_array_equal(vt, g["vt"][lo:hi])
        want_tag = 3 if g["mode"][c] else 2
        assert set(g["adv_t"][lo:hi]) == {want_tag}


def test_gae_fragments_of_the_recurrent_dataset():
    """mode 2: v_pred / next_v_pred are Python floats (reference ppo.py:98-107), so ppo.py:36-47
    runs in f64 throughout -- the product gamma * nonterminal * next_v included."""
    g = np.load(os.path.join(GOLDEN, "gae_recurrent.npz"))
    differs = 0
    for c in range(len(g["gamma"])):
        lo, hi = g["off"][c], g["off"][c + 1]
        args = (g["reward"][lo:hi], g["v"][lo:hi], g["nv"][lo:hi], g["nonterm"][lo:hi],
                g["gamma"][c], g["lambd"][c])
        adv, vt = oracle.gae_fragment(*args, 2)
        np.testing.assert_array_equal(adv, g["adv"][lo:hi])
        np.testing.assert_array_equal(vt, g["vt"][lo:hi])
        assert set(g["adv_t"][lo:hi]) <= {1, 3}      # Python float / np.float64: f64 either way
        differs += int(not np.array_equal(oracle.gae_fragment(*args, 1)[0], adv))
    assert differs > 0      # (the f32-rounded product of mode 1 is a different number)


def test_a2c_returns():
    g = np.load(os.path.join(GOLDEN, "a2c_returns.npz"))
    for c in range(4):
        T, N, use_gae = (int(x) for x in g["c%d_meta" % c])
        gamma, tau = (float(x) for x in g["c%d_hyper" % c])
        ret = oracle.a2c_returns(g["c%d_rewards" % c], g["c%d_masks" % c],
                                 g["c%d_value_preds" % c], g["c%d_next_value" % c],
                                 gamma, tau, use_gae)
        want = g["c%d_returns" % c]
        if use_gae:
            np.testing.assert_array_equal(ret[:T], want[:T])
        else:
            np.testing.assert_array_equal(ret, want)


def test_c51_loss_oracle_matches_reference_golden():
    """orc_c51_loss (plain C) against the vectors recorded from the reference's own
    functions: projected target bit-exact (same float32 accumulation order as the CPU
    scatter_add_), loss / gradient / KL / Q(s, a) to float32 rounding of log and sums."""
    g = np.load(os.path.join(GOLDEN, "c51_loss.npz"))
    for ci in range(int(g["n_cases"])):
        k = lambda name: g["k%d_%s" % (ci, name)]
        double, weighted, mean = (bool(v) for v in k("flags"))
        out = oracle.c51_loss(k("q_dist"), k("action"), k("next_dist"),
                              k("next_sel") if double else None, k("z"), k("reward"),
                              k("discount"), k("terminal"), k("weights") if weighted else None,
                              mean)
        np.testing.assert_array_equal(out["target"], k("target"))
        np.testing.assert_allclose(out["loss"], float(k("loss")), rtol=2e-6)
        np.testing.assert_allclose(out["grad"], k("grad"), rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(out["delta"], k("delta"), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(out["qsa"], k("qsa"), rtol=2e-6, atol=1e-7)


def test_dqn_td_loss_oracle_matches_reference_golden():
    """orc_dqn_td_loss against the reference's compute_(weighted_)value_loss and
    (Double-)DQN targets: y and t bit-exact, loss / gradient to sum rounding."""
    g = np.load(os.path.join(GOLDEN, "td_loss.npz"))
    for ci in range(int(g["n_cases"])):
        k = lambda name: g["k%d_%s" % (ci, name)]
        double, clip, mean, weighted = (bool(v) for v in k("flags"))
        out = oracle.dqn_td_loss(k("q"), k("action"), k("tq"), k("nq") if double else None,
                                 k("reward"), k("discount"), k("terminal"),
                                 k("weights") if weighted else None, clip, mean)
        np.testing.assert_array_equal(out["y"], k("y"))
        np.testing.assert_array_equal(out["t"], k("t"))
        np.testing.assert_allclose(out["loss"], float(k("loss")), rtol=2e-6)
        np.testing.assert_allclose(out["grad"], k("grad"), rtol=2e-6, atol=1e-9)


def test_sample_n_k_indices_and_stream_position():
    """pfrl_amd.utils.random.sample_n_k against the reference's sample_n_k
    (pfrl/utils/random.py:4-28) on 11 (n, k) cases: the same indices from the same seed,
    and the same amount of the global NumPy stream consumed (the next draw matches)."""
    from pfrl_amd.utils.random import sample_n_k

    g = np.load(os.path.join(GOLDEN, "sample_n_k.npz"))
    for i in range(len(g["n"])):
        np.random.seed(int(g["seed"][i]))
        idx = sample_n_k(int(g["n"][i]), int(g["k"][i]))
        tail = np.random.random_sample()
        want = g["idx"][g["off"][i]:g["off"][i + 1]]
        np.testing.assert_array_equal(np.asarray(idx, dtype=np.int64), want)
        assert tail == g["tail"][i]
        assert len(set(int(j) for j in idx)) == int(g["k"][i])      # distinct
    with pytest.raises(ValueError):
        sample_n_k(3, 4)
    with pytest.raises(ValueError):
        sample_n_k(3, -1)
