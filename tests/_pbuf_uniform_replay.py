"""Replay of tests/golden/pbufmix_trace_*.npz (recorded from the reference's PrioritizedBuffer
with uniform_ratio > 0 and / or wait_priority_after_sampling=False, make_golden.pbuf_uniform_trace)
on any buffer with the reference's interface.  The replay re-seeds the GLOBAL NumPy stream as the
recording did: binomial split, sample_n_k and the prioritized draws must consume it in the
reference's order for the indices to come out."""
import numpy as np


def _scalar(v, t):
    return {0: None, 1: float(v), 2: np.float32(v), 3: np.float64(v)}[int(t)]


def _tag(x):
    if isinstance(x, np.float32):
        return 2
    if isinstance(x, np.float64):
        return 3
    assert isinstance(x, (float, int)), type(x)
    return 1


def replay(g, make_buffer, root_stats):
    """``make_buffer(capacity, wait)`` -> buffer; ``root_stats(buf)`` -> ((sum, tag), (min, tag),
    (max_priority, tag)) or None when empty.  Returns the buffer for the caller's tree dump."""
    seed, cap, batch, wait = (int(x) for x in g["meta"])
    ur = float(g["uniform_ratio"])
    np.random.seed(seed)
    buf = make_buffer(None if cap < 0 else cap, bool(wait))
    ops = g["op_kind"]
    i_app = i_idx = i_set = i_smp = 0
    step = 0          # index into the per-op statistics (a set_last_priority shares its sample's)
    payload = 0
    k = 0
    while k < len(ops):
        op = int(ops[k])
        if op == 2:
            sampled, probs, min_prob = buf.sample(batch, uniform_ratio=ur)
            n = int(g["n_sampled"][i_smp])
            assert len(sampled) == n == batch
            want_idx = g["idx"][i_idx:i_idx + n]
            np.testing.assert_array_equal(np.asarray(buf.sampled_indices), want_idx, err_msg="sample %d" % i_smp)
            assert [sampled[j] for j in range(n)] == [int(buf.data[i]) for i in want_idx]
            for j, p in enumerate(probs):
                assert float(p) == g["prob_v"][i_idx + j], (i_smp, j)
                assert _tag(p) == g["prob_t"][i_idx + j], (i_smp, j, type(p))
            assert float(min_prob) == g["min_prob_v"][i_smp] and _tag(min_prob) == g["min_prob_t"][i_smp]
            i_idx += n
            i_smp += 1
            if k + 1 < len(ops) and int(ops[k + 1]) == 3:
                newp = [_scalar(v, t) for v, t in zip(g["set_v"][i_set:i_set + n], g["set_t"][i_set:i_set + n])]
                buf.set_last_priority(newp)
                i_set += n
                k += 1
        elif op == 4:
            buf.popleft()
        else:
            p = _scalar(g["app_v"][i_app], g["app_t"][i_app]) if op == 1 else None
            buf.append(payload, priority=p)
            payload += 1
            i_app += 1
        st = root_stats(buf)
        assert len(buf) == int(g["length"][step])
        if st is not None:
            assert st[0] == (g["sum_v"][step], g["sum_t"][step]), ("sum", step, st[0])
            assert st[1] == (g["min_v"][step], g["min_t"][step]), ("min", step)
            assert st[2] == (g["maxp_v"][step], g["maxp_t"][step]), ("max_priority", step)
        step += 1
        k += 1
    assert i_smp == len(g["n_sampled"]) and step == len(g["length"])
    return buf
