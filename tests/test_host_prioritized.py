"""The host-side prioritized buffer (``gpu=None`` plumbing path) against the traces recorded
from the reference (tests/golden/pbuf_trace_*.npz): scalar VALUES AND TYPES of every removed
priority, of the root sum / min and of max_priority after every operation, the frame bounds, the
sampled indices on the reference's own NumPy stream, and the full final trees."""
import glob
import os

import numpy as np
import pytest

from pfrl_amd.collections.host_prioritized import HostPrioritizedBuffer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scalar(v, t):
    return {0: None, 1: float(v), 2: np.float32(v), 3: np.float64(v)}[int(t)]


def _tag(x):
    if x is None:
        return 0
    if isinstance(x, np.float32):
        return 2
    if isinstance(x, np.float64):
        return 3
    assert isinstance(x, (float, int)), type(x)
    return 1


def _typed(x):
    return float(x), _tag(x)


def _check_levels(tq, values, tags):
    """Level-order dump as make_golden.flat_dump writes it: leaves first, absent nodes as (0, 0)."""
    f = tq.frame
    off = 0
    for l in range(f.log2_size + 1):
        n = f.size >> l
        q0 = (f.base - f.origin[l]) >> l
        for j in range(n):
            node = tq.levels[l].get(q0 + j)
            want = (values[off + j], tags[off + j])
            assert ((0.0, 0) if node is None else _typed(node)) == want, (l, j)
        off += n
    assert off == len(values)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "pbuf_trace_*.npz"))),
                         ids=os.path.basename)
def test_host_prioritized_buffer_follows_reference_trace(path):
    g = np.load(path)
    seed, cap = int(g["meta"][0]), int(g["meta"][1])
    np.random.seed(seed)
    buf = HostPrioritizedBuffer(None if cap < 0 else cap)
    ia = iu = ismp = 0
    payload = 0
    for k, (kind, n) in enumerate(zip(g["op_kind"], g["op_n"])):
        n = int(n)
        if kind in (0, 1):
            buf.append(payload, priority=_scalar(g["app_v"][ia], g["app_t"][ia]))
            ia += 1
            payload += 1
        elif kind == 4:
            buf.popleft()
        else:
            total = buf.priority_sums.sum()
            removed = {}
            real = buf.priority_sums.prioritized_sample

            def spy(n_, remove, _real=real, _out=removed):
                ixs, vals = _real(n_, remove)
                _out["vals"] = vals
                return ixs, vals

            buf.priority_sums.prioritized_sample = spy
            sampled, probs, min_prob = buf.sample(n)
            del buf.priority_sums.prioritized_sample
            sl = slice(iu, iu + n)
            assert buf.sampled_indices == list(g["idx"][sl]), k
            assert sampled == [buf.data[i] for i in g["idx"][sl]]
            assert [_typed(v) for v in removed["vals"]] == list(zip(g["pri_v"][sl], g["pri_t"][sl]))
            np.testing.assert_array_equal(np.asarray(probs, dtype=np.float64), g["prob"][sl])
            assert _typed(total) == (g["total_v"][ismp], g["total_t"][ismp])
            assert float(min_prob) == g["min_prob"][ismp]
            with pytest.raises(AssertionError):
                buf.sample(n)                         # priorities of the last sample are pending
            buf.set_last_priority([_scalar(v, t) for v, t in zip(g["set_v"][sl], g["set_t"][sl])])
            iu += n
            ismp += 1
        assert len(buf) == g["length"][k]
        if len(buf):
            assert _typed(buf.priority_sums.sum()) == (g["sum_v"][k], g["sum_t"][k]), k
            assert _typed(buf.priority_mins.min()) == (g["min_v"][k], g["min_t"][k]), k
            assert buf.priority_sums.bounds == (g["ixl"][k], g["ixr"][k]), k
            assert buf.priority_mins.bounds == buf.priority_sums.bounds
        assert _typed(buf.max_priority) == (g["maxp_v"][k], g["maxp_t"][k]), k
    _check_levels(buf.priority_sums, g["final_sum_v"], g["final_sum_t"])
    _check_levels(buf.priority_mins, g["final_min_v"], g["final_min_t"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "pbufmix_trace_*.npz"))),
                         ids=os.path.basename)
def test_host_prioritized_buffer_uniform_ratio_and_no_wait_follow_reference_trace(path):
    """uniform_ratio > 0 (binomial split + SumTreeQueue.uniform_sample) and
    wait_priority_after_sampling=False (priorities written back after the draws): indices on
    the reference's NumPy stream, probabilities and min_prob with their NEP-50 types, root
    statistics after every operation, the final trees."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _pbuf_uniform_replay import replay

    def stats(buf):
        if len(buf) == 0:
            return None
        return (_typed(buf.priority_sums.sum()), _typed(buf.priority_mins.min()), _typed(buf.max_priority))

    g = np.load(path)
    buf = replay(g, lambda cap, wait: HostPrioritizedBuffer(cap, wait_priority_after_sampling=wait), stats)
    _check_levels(buf.priority_sums, g["final_sum_v"], g["final_sum_t"])
    _check_levels(buf.priority_mins, g["final_min_v"], g["final_min_t"])


def test_host_prioritized_buffer_uniform_mixture_and_misuse():
    np.random.seed(3)
    buf = HostPrioritizedBuffer(capacity=8)
    for i in range(12):
        buf.append(i, priority=None if i % 3 else 0.5 + i)
    assert len(buf) == 8 and list(buf.data) == list(range(4, 12))
    with pytest.raises(AssertionError):
        buf.set_last_priority([1.0])                  # nothing sampled yet
    sampled, probs, min_prob = buf.sample(4, uniform_ratio=0.5)
    assert len(set(sampled)) == 4 and all(p > 0 for p in probs) and min_prob > 0
    with pytest.raises(AssertionError):
        buf.set_last_priority([1.0, 0.0, 1.0, 1.0])   # priorities must be positive
    buf.set_last_priority([2.0, 3.0, 4.0, 5.0])
    assert buf.max_priority == 5.0 and not buf.flag_wait_priority   # appended priorities do not raise it
    total = sum(buf.priority_sums.levels[0].values())
    assert abs(buf.priority_sums.sum() - total) < 1e-9
    while len(buf):
        buf.popleft()
    assert buf.priority_sums.sum() == 0.0 and buf.priority_mins.min() == np.inf
    assert not any(buf.priority_sums.levels) and not any(buf.priority_mins.levels)


@pytest.mark.parametrize("name,prioritized,num_steps,double", [
    ("dqn_uniform_n1", False, 1, False), ("ddqn_per_n3", True, 3, True)])
def test_dqn_family_host_mode_matches_reference_traces(tmp_path, name, prioritized, num_steps,
                                                       double):
    """DQN / DoubleDQN created with ``gpu=-1`` (host replay, host priority trees) against the
    traces the reference recorded on the CPU: every action, every sampled minibatch (lengths and
    reward sums), every loss, the trained parameters, statistics and -- with prioritized replay --
    the final tree sum and max_priority."""
    import torch

    import pfrl_amd
    from pfrl_amd import agents, experiments, explorers, replay_buffers
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv
    from pfrl_amd.q_functions import DiscreteActionValueHead

    g = np.load(os.path.join(GOLDEN, "agent_trace_%s.npz" % name))
    pfrl_amd.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(4, seed=3, frame_shape=(12, 12), p_done=0.04)
    torch.manual_seed(1234)
    q = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU(),
                            torch.nn.Linear(32, 6), DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2)
    if prioritized:
        rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                      num_steps=num_steps,
                                                      normalize_by_max="memory")
    else:
        rbuf = replay_buffers.ReplayBuffer(200, num_steps=num_steps)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    cls = agents.DoubleDQN if double else agents.DQN
    ag = cls(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40, minibatch_size=8,
             update_interval=4, target_update_interval=60,
             phi=lambda x: np.asarray(x, dtype=np.float32) / 255, batch_accumulator="sum")
    assert not rbuf.is_device
    actions, losses, sampled = [], [], []
    orig_act, orig_update = ag.batch_act, ag.update

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    def spy_update(exps, errors_out=None):
        sampled.append([[float(np.asarray(t["reward"])) for t in e] for e in exps])
        orig_update(exps, errors_out)
        losses.append(float(ag.loss_record.values()[-1]))

    ag.batch_act = spy_act
    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, 640, str(tmp_path))
    np.testing.assert_array_equal(np.asarray(actions), g["actions"])
    np.testing.assert_array_equal(np.asarray([[len(e) for e in s] for s in sampled]),
                                  g["sampled_len"])
    np.testing.assert_allclose([sum(r for e in s for r in e) for s in sampled],
                               g["sampled_reward_sum"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.asarray(losses), g["losses"], rtol=1e-5, atol=1e-6)
    params = np.concatenate([p.detach().numpy().ravel() for p in q.parameters()])
    np.testing.assert_allclose(params, g["final_params"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([float(v) for _, v in ag.get_statistics()], g["stats"], rtol=1e-5,
                               atol=1e-6)
    if prioritized:
        np.testing.assert_allclose(float(rbuf.memory.priority_sums.sum()),
                                   float(g["final_tree_sum"]), rtol=1e-6)
        np.testing.assert_allclose(float(rbuf.memory.max_priority),
                                   float(g["final_max_priority"]), rtol=1e-6)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "per_trace_*.npz"))),
                         ids=os.path.basename)
def test_host_prioritized_replay_buffer_follows_reference_trace(path):
    """``PrioritizedReplayBuffer`` used without a GPU, on top of the n-step windows: sampled
    entries, probabilities, importance weights and the beta schedule, the typed priorities that
    ``update_errors`` writes (f32 / Python-float errors mixed), the typed root sum / min /
    max_priority and the frame bounds after every operation, and the full trees at the
    checkpoints the fixture holds."""
    from pfrl_amd.replay_buffers import PrioritizedReplayBuffer

    g = np.load(path)
    seed, cap, n_steps, batch, n_envs = (int(x) for x in g["meta"])
    alpha, beta0, betasteps, eps = (float(x) for x in g["hyper"])
    norm = {0: False, 1: True, 2: "memory"}[int(g["normalize_by_max"])]
    np.random.seed(seed)
    rbuf = PrioritizedReplayBuffer(capacity=None if cap < 0 else cap, alpha=alpha, beta0=beta0,
                                   betasteps=betasteps, normalize_by_max=norm, num_steps=n_steps)
    assert eps == rbuf.eps
    tid = iu = ismp = idump = 0
    for k, (kind, a, b) in enumerate(zip(g["op_kind"], g["op_a"], g["op_b"])):
        if kind == 0:
            rbuf.append(state=tid, action=0, reward=0.0, next_state=tid + 1,
                        is_state_terminal=bool(b), env_id=int(a), tid=tid)
            tid += 1
        elif kind == 1:
            rbuf.stop_current_episode(env_id=int(a))
        else:
            mem = rbuf.memory
            sl = slice(iu, iu + batch)
            assert _typed(mem.priority_sums.sum()) == (g["smp_total_v"][ismp], g["smp_total_t"][ismp])
            assert rbuf.beta == g["beta"][ismp]
            sampled = rbuf.sample(batch)
            assert mem.sampled_indices == list(g["idx"][sl]), k
            for j, entry in enumerate(sampled):
                want = g["entry_tids"][(iu + j) * n_steps:(iu + j + 1) * n_steps]
                assert [tr["tid"] for tr in entry] == [int(x) for x in want if x >= 0]
            np.testing.assert_allclose([e[0]["weight"] for e in sampled], g["weight"][sl],
                                       rtol=1e-12)
            errors = [float(e) if is_py else np.float32(e)
                      for e, is_py in zip(g["err"][sl], g["err_is_py"][sl])]
            written = {}
            real = mem.set_last_priority
            mem.set_last_priority = lambda pri, _w=written, _r=real: (_w.update(p=list(pri)),
                                                                      _r(pri))[1]
            rbuf.update_errors(errors)
            del mem.set_last_priority
            assert [_typed(p) for p in written["p"]] == list(zip(g["new_pri_v"][sl],
                                                                 g["new_pri_t"][sl])), k
            iu += batch
            ismp += 1
        mem = rbuf.memory
        assert len(mem) == g["length"][k], k
        if len(mem):
            assert _typed(mem.priority_sums.sum()) == (g["sum_v"][k], g["sum_t"][k]), k
            assert _typed(mem.priority_mins.min()) == (g["min_v"][k], g["min_t"][k]), k
            assert mem.priority_sums.bounds == (g["ixl"][k], g["ixr"][k]), k
        assert _typed(mem.max_priority) == (g["maxp_v"][k], g["maxp_t"][k]), k
        if idump < len(g["dump_op"]) and g["dump_op"][idump] == k:
            lo, hi = g["dump_off"][idump], g["dump_off"][idump + 1]
            if hi > lo:
                _check_levels(mem.priority_sums, g["dump_sum_v"][lo:hi], g["dump_sum_t"][lo:hi])
                _check_levels(mem.priority_mins, g["dump_min_v"][lo:hi], g["dump_min_t"][lo:hi])
            idump += 1
    assert idump == len(g["dump_op"]) and ismp == len(g["beta"])
