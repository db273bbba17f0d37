#!/usr/bin/env python
"""Record golden vectors by running the REFERENCE (pfnet/pfrl) itself.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

The reference cannot travel to the GPU box, so the vectors are committed as
small ``.npz`` fixtures next to this script.  They pin the CPU oracle
(tests/test_oracle_golden.py) and, through it and directly, the HIP path.

Scalars that live inside the reference's priority trees are recorded as
(float64 value, type tag) pairs: 0 absent, 1 Python float, 2 np.float32,
3 np.float64 -- see oracle/pfrl_oracle.c.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "_gymshim"))
sys.path.insert(0, os.environ.get("PFRL_REFERENCE", "/root/reference"))

import random  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import pfrl  # noqa: E402
from pfrl.collections.prioritized import PrioritizedBuffer  # noqa: E402
from pfrl.replay_buffer import batch_experiences  # noqa: E402
from pfrl.replay_buffers import PrioritizedReplayBuffer, ReplayBuffer  # noqa: E402


def tag(x):
    if x is None:
        return 0
    if isinstance(x, np.float32):
        return 2
    if isinstance(x, np.float64):
        return 3
    if isinstance(x, (float, int)):
        return 1
    raise TypeError(type(x))


def val(x):
    return 0.0 if x is None else float(x)


def dump_tree(tq):
    """Level-order dump of a TreeQueue: dict width -> (values, tags)."""
    out = {}
    if tq.length == 0:
        return out
    ixl, ixr = tq.bounds
    size = ixr - ixl

    def rec(node, lo, hi, width, vs, ts):
        if hi - lo == width:
            j = (lo - ixl) // width
            if node:
                vs[j] = val(node[2])
                ts[j] = tag(node[2])
            return
        c = (lo + hi) // 2
        left = node[0] if node else []
        right = node[1] if node else []
        rec(left if left is not None else [], lo, c, width, vs, ts)
        rec(right if right is not None else [], c, hi, width, vs, ts)

    w = 1
    while w <= size:
        vs = np.zeros(size // w, dtype=np.float64)
        ts = np.zeros(size // w, dtype=np.int32)
        rec(tq.root, ixl, ixr, w, vs, ts)
        out[w] = (vs, ts)
        w *= 2
    return out


def flat_dump(tq):
    d = dump_tree(tq)
    if not d:
        return np.zeros(0), np.zeros(0, dtype=np.int32)
    vs = np.concatenate([d[w][0] for w in sorted(d)])
    ts = np.concatenate([d[w][1] for w in sorted(d)])
    return vs, ts


# --------------------------------------------------------------------------
# A. PrioritizedReplayBuffer traces
# --------------------------------------------------------------------------
def per_trace(name, seed, capacity, num_steps, n_ops, batch, err_kind, normalize_by_max,
              alpha=0.6, beta0=0.4, betasteps=200, n_envs=3, dump_every=25):
    np.random.seed(seed)
    rs = np.random.RandomState(seed + 1000)  # script randomness, separate stream
    rbuf = PrioritizedReplayBuffer(
        capacity=capacity, alpha=alpha, beta0=beta0, betasteps=betasteps,
        normalize_by_max=normalize_by_max, num_steps=num_steps,
    )
    captured = {}
    orig_set = rbuf.memory.set_last_priority

    def spy_set(priority):
        captured["pri"] = list(priority)
        return orig_set(priority)

    rbuf.memory.set_last_priority = spy_set

    ops = []  # (kind, a, b)   kind: 0 append(env, terminal) 1 stop(env) 2 sample(n)+update
    rec = dict(
        op_kind=[], op_a=[], op_b=[], length=[], sum_v=[], sum_t=[], min_v=[], min_t=[],
        maxp_v=[], maxp_t=[], ixl=[], ixr=[],
        u01=[], idx=[], pri_v=[], pri_t=[], weight=[], entry_tids=[], entry_len=[],
        err=[], err_is_py=[], new_pri_v=[], new_pri_t=[], beta=[],
        dump_op=[], dump_sum_v=[], dump_sum_t=[], dump_min_v=[], dump_min_t=[], dump_off=[0],
        smp_total_v=[], smp_total_t=[], smp_min_prob=[], prob=[],
    )
    tid = 0
    for k in range(n_ops):
        r = rs.rand()
        can_sample = len(rbuf) >= batch
        if can_sample and r < 0.3:
            kind = 2
        elif r < 0.38:
            kind = 1
        else:
            kind = 0
        if kind == 0:
            env = int(rs.randint(n_envs))
            term = bool(rs.rand() < 0.15)
            rbuf.append(state=tid, action=0, reward=float(rs.randn()), next_state=tid + 1,
                        is_state_terminal=term, env_id=env, tid=tid)
            rec["op_kind"].append(0); rec["op_a"].append(env); rec["op_b"].append(int(term))
            tid += 1
        elif kind == 1:
            env = int(rs.randint(n_envs))
            rbuf.stop_current_episode(env_id=env)
            rec["op_kind"].append(1); rec["op_a"].append(env); rec["op_b"].append(0)
        else:
            st = np.random.get_state()
            total_before = rbuf.memory.priority_sums.sum()
            beta_before = rbuf.beta
            # spy on the collections-level sample to get indices/priorities/probs
            orig = rbuf.memory._sample_indices_and_probabilities
            got = {}

            def spy(n, uniform_ratio, _orig=orig, _got=got):
                # re-implement the call to read sampled priorities too
                res = _orig(n, uniform_ratio)
                _got["res"] = res
                return res

            rbuf.memory._sample_indices_and_probabilities = spy
            sampled = rbuf.sample(batch)
            rbuf.memory._sample_indices_and_probabilities = orig
            indices, probs, min_prob = got["res"]
            st_after = np.random.get_state()
            np.random.set_state(st)
            u = np.random.random_sample(batch)
            st_chk = np.random.get_state()
            assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
                       for a, b in zip(st_after, st_chk)), "rng stream mismatch"
            rec["u01"].extend(u.tolist())
            rec["idx"].extend(int(i) for i in indices)
            rec["prob"].extend(float(p) for p in probs)
            rec["smp_total_v"].append(val(total_before)); rec["smp_total_t"].append(tag(total_before))
            rec["smp_min_prob"].append(float(min_prob))
            rec["beta"].append(beta_before)
            for e in sampled:
                rec["weight"].append(float(e[0]["weight"]))
                tids = [tr["tid"] for tr in e]
                rec["entry_len"].append(len(tids))
                rec["entry_tids"].extend(tids + [-1] * (num_steps - len(tids)))
            # errors
            if err_kind == "f32":
                errs = [np.float32(x) for x in (rs.rand(batch) * 1.6).astype(np.float32)]
                if rs.rand() < 0.5:
                    errs[int(rs.randint(batch))] = np.float32(0.0)
            elif err_kind == "py":
                errs = [float(x) for x in rs.rand(batch) * 1.6]
            else:  # mixed
                errs = [np.float32(x) if rs.rand() < 0.7 else float(x)
                        for x in rs.rand(batch) * 1.6]
            rbuf.update_errors(errs)
            rec["err"].extend(float(e) for e in errs)
            rec["err_is_py"].extend(int(not isinstance(e, np.float32)) for e in errs)
            rec["new_pri_v"].extend(val(p) for p in captured["pri"])
            rec["new_pri_t"].extend(tag(p) for p in captured["pri"])
            # sampled priorities (removed values) are not returned by the
            # reference API; recover them from probs*total is inexact, so read
            # them through a second spy below (see pri_v note)
            rec["op_kind"].append(2); rec["op_a"].append(batch); rec["op_b"].append(0)
        mem = rbuf.memory
        rec["length"].append(len(mem))
        s = mem.priority_sums.sum() if len(mem) else 0.0
        m = mem.priority_mins.min() if len(mem) else float("inf")
        rec["sum_v"].append(val(s)); rec["sum_t"].append(tag(s))
        rec["min_v"].append(val(m)); rec["min_t"].append(1 if not len(mem) else tag(m))
        rec["maxp_v"].append(val(mem.max_priority)); rec["maxp_t"].append(tag(mem.max_priority))
        if len(mem):
            rec["ixl"].append(mem.priority_sums.bounds[0]); rec["ixr"].append(mem.priority_sums.bounds[1])
        else:
            rec["ixl"].append(0); rec["ixr"].append(0)
        if (k % dump_every == dump_every - 1) or k == n_ops - 1:
            sv, st_ = flat_dump(mem.priority_sums)
            mv, mt = flat_dump(mem.priority_mins)
            rec["dump_op"].append(k)
            rec["dump_sum_v"].append(sv); rec["dump_sum_t"].append(st_)
            rec["dump_min_v"].append(mv); rec["dump_min_t"].append(mt)
            rec["dump_off"].append(rec["dump_off"][-1] + len(sv))
    out = {}
    for key, v in rec.items():
        if key.startswith("dump_") and key not in ("dump_op", "dump_off"):
            out[key] = np.concatenate(v) if v else np.zeros(0)
        else:
            out[key] = np.asarray(v)
    out["meta"] = np.array([seed, -1 if capacity is None else capacity, num_steps, batch, n_envs])
    out["hyper"] = np.array([alpha, beta0, betasteps, 0.01], dtype=np.float64)
    out["normalize_by_max"] = np.array(
        {True: 1, "batch": 1, "memory": 2, False: 0}[normalize_by_max])
    np.savez_compressed(os.path.join(HERE, "per_trace_%s.npz" % name), **out)
    print("per_trace", name, "ops", n_ops, "final len", len(rbuf))


# --------------------------------------------------------------------------
# A'. collections-level PrioritizedBuffer trace (records removed priorities)
# --------------------------------------------------------------------------
def pbuf_trace(name, seed, capacity, n_ops, batch):
    np.random.seed(seed)
    rs = np.random.RandomState(seed + 77)
    buf = PrioritizedBuffer(capacity=capacity)
    rec = dict(op_kind=[], op_n=[], u01=[], idx=[], pri_v=[], pri_t=[], prob=[], min_prob=[],
               total_v=[], total_t=[], set_v=[], set_t=[], app_v=[], app_t=[],
               sum_v=[], sum_t=[], min_v=[], min_t=[], maxp_v=[], maxp_t=[], length=[],
               ixl=[], ixr=[])
    payload = 0
    for k in range(n_ops):
        r = rs.rand()
        if len(buf) >= batch and r < 0.3:
            st = np.random.get_state()
            total = buf.priority_sums.sum()
            # replicate PrioritizedBuffer.sample but keep the removed priorities
            tq = buf.priority_sums
            got = {}
            orig = tq.prioritized_sample

            def spy(n, remove, _orig=orig, _got=got):
                ixs, vals = _orig(n, remove)
                _got["vals"] = vals
                return ixs, vals

            tq.prioritized_sample = spy
            sampled, probs, min_prob = buf.sample(batch)
            tq.prioritized_sample = orig
            np.random.set_state(st)
            u = np.random.random_sample(batch)
            rec["op_kind"].append(2); rec["op_n"].append(batch)
            rec["u01"].extend(u.tolist())
            rec["idx"].extend(int(i) for i in buf.sampled_indices)
            rec["pri_v"].extend(val(v) for v in got["vals"])
            rec["pri_t"].extend(tag(v) for v in got["vals"])
            rec["prob"].extend(float(p) for p in probs)
            rec["min_prob"].append(float(min_prob))
            rec["total_v"].append(val(total)); rec["total_t"].append(tag(total))
            # new priorities: mixture of types
            newp = []
            for _ in range(batch):
                c = rs.rand()
                x = rs.rand() * 3 + 1e-3
                if c < 0.5:
                    newp.append(np.float32(x))
                elif c < 0.9:
                    newp.append(float(x))
                else:
                    newp.append(np.float64(x))
            buf.set_last_priority(newp)
            rec["set_v"].extend(val(p) for p in newp)
            rec["set_t"].extend(tag(p) for p in newp)
        elif len(buf) > 0 and r < 0.36:
            buf.popleft()
            rec["op_kind"].append(4); rec["op_n"].append(0)
        else:
            c = rs.rand()
            if c < 0.6:
                p = None
            elif c < 0.8:
                p = float(rs.rand() * 2 + 0.01)
            else:
                p = np.float32(rs.rand() * 2 + 0.01)
            buf.append(payload, priority=p)
            payload += 1
            rec["op_kind"].append(0 if p is None else 1); rec["op_n"].append(0)
            rec["app_v"].append(val(p)); rec["app_t"].append(tag(p))
        s = buf.priority_sums.sum() if len(buf) else 0.0
        m = buf.priority_mins.min() if len(buf) else float("inf")
        rec["sum_v"].append(val(s)); rec["sum_t"].append(tag(s))
        rec["min_v"].append(val(m)); rec["min_t"].append(1 if not len(buf) else tag(m))
        rec["maxp_v"].append(val(buf.max_priority)); rec["maxp_t"].append(tag(buf.max_priority))
        rec["length"].append(len(buf))
        if len(buf):
            rec["ixl"].append(buf.priority_sums.bounds[0]); rec["ixr"].append(buf.priority_sums.bounds[1])
        else:
            rec["ixl"].append(0); rec["ixr"].append(0)
    out = {k2: np.asarray(v) for k2, v in rec.items()}
    sv, st_ = flat_dump(buf.priority_sums)
    mv, mt = flat_dump(buf.priority_mins)
    out.update(final_sum_v=sv, final_sum_t=st_, final_min_v=mv, final_min_t=mt)
    out["meta"] = np.array([seed, -1 if capacity is None else capacity, batch])
    np.savez_compressed(os.path.join(HERE, "pbuf_trace_%s.npz" % name), **out)
    print("pbuf_trace", name, "final len", len(buf))


# --------------------------------------------------------------------------
# A''. PrioritizedBuffer with uniform_ratio > 0 and / or wait_priority_after_sampling=False
#      (prioritized.py:56-84, 278-312): the replay re-seeds the GLOBAL NumPy stream and performs
#      the same calls; binomial, sample_n_k and the prioritized draws consume it in that order
# --------------------------------------------------------------------------
def pbuf_uniform_trace(name, seed, capacity, n_ops, batch, uniform_ratio, wait):
    np.random.seed(seed)
    rs = np.random.RandomState(seed + 177)
    buf = PrioritizedBuffer(capacity=capacity, wait_priority_after_sampling=wait)
    rec = dict(op_kind=[], idx=[], prob_v=[], prob_t=[], min_prob_v=[], min_prob_t=[], set_v=[], set_t=[],
               app_v=[], app_t=[], sum_v=[], sum_t=[], min_v=[], min_t=[], maxp_v=[], maxp_t=[],
               length=[], n_sampled=[])
    payload = 0
    for k in range(n_ops):
        r = rs.rand()
        if len(buf) >= batch and r < 0.3:
            sampled, probs, min_prob = buf.sample(batch, uniform_ratio=uniform_ratio)
            rec["op_kind"].append(2)
            rec["n_sampled"].append(len(sampled))
            rec["idx"].extend(int(i) for i in buf.sampled_indices)
            rec["prob_v"].extend(val(p) for p in probs); rec["prob_t"].extend(tag(p) for p in probs)
            rec["min_prob_v"].append(val(min_prob)); rec["min_prob_t"].append(tag(min_prob))
            # with wait=False the priorities may, but need not, be set afterwards
            if wait or rs.rand() < 0.5:
                newp = []
                for _ in range(batch):
                    c = rs.rand()
                    x = rs.rand() * 3 + 1e-3
                    newp.append(np.float32(x) if c < 0.5 else (float(x) if c < 0.9 else np.float64(x)))
                buf.set_last_priority(newp)
                rec["op_kind"].append(3)
                rec["set_v"].extend(val(p) for p in newp); rec["set_t"].extend(tag(p) for p in newp)
        elif len(buf) > 0 and r < 0.36:
            buf.popleft()
            rec["op_kind"].append(4)
        else:
            c = rs.rand()
            p = None if c < 0.6 else (float(rs.rand() * 2 + 0.01) if c < 0.8 else np.float32(rs.rand() * 2 + 0.01))
            buf.append(payload, priority=p)
            payload += 1
            rec["op_kind"].append(0 if p is None else 1)
            rec["app_v"].append(val(p)); rec["app_t"].append(tag(p))
        s = buf.priority_sums.sum() if len(buf) else 0.0
        m = buf.priority_mins.min() if len(buf) else float("inf")
        rec["sum_v"].append(val(s)); rec["sum_t"].append(tag(s))
        rec["min_v"].append(val(m)); rec["min_t"].append(1 if not len(buf) else tag(m))
        rec["maxp_v"].append(val(buf.max_priority)); rec["maxp_t"].append(tag(buf.max_priority))
        rec["length"].append(len(buf))
    out = {k2: np.asarray(v) for k2, v in rec.items()}
    sv, st_ = flat_dump(buf.priority_sums)
    mv, mt = flat_dump(buf.priority_mins)
    out.update(final_sum_v=sv, final_sum_t=st_, final_min_v=mv, final_min_t=mt)
    out["meta"] = np.array([seed, -1 if capacity is None else capacity, batch, int(wait)])
    out["uniform_ratio"] = np.array(uniform_ratio, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "pbufmix_trace_%s.npz" % name), **out)
    print("pbuf_uniform_trace", name, "final len", len(buf), "samples", len(rec["n_sampled"]))


def prioritized_episodic_uniform_traces():
    # (a glob of their own: the CPU test of the plain traces runs on the C oracle's tree)
    prioritized_episodic_trace("cap30_r03", 43, 30, 500, 3, uniform_ratio=0.3,
                               prefix="prioritized_episodic_uniform_trace")
    prioritized_episodic_trace("unbounded_r06_memory", 44, None, 300, 2, normalize_by_max="memory",
                               uniform_ratio=0.6, prefix="prioritized_episodic_uniform_trace")


def pbuf_uniform_traces():
    pbuf_uniform_trace("r03_cap64", 21, 64, 900, 8, 0.3, True)
    pbuf_uniform_trace("r05_cap1000", 22, 1000, 1400, 32, 0.5, True)
    pbuf_uniform_trace("r10_unbounded", 23, None, 600, 6, 1.0, True)
    pbuf_uniform_trace("r00_nowait_cap100", 24, 100, 900, 8, 0.0, False)
    pbuf_uniform_trace("r04_nowait_cap300", 25, 300, 1500, 16, 0.4, False)


# --------------------------------------------------------------------------
# B. uniform ReplayBuffer n-step traces + batch_experiences
# --------------------------------------------------------------------------
def replay_trace(name, seed, capacity, num_steps, n_ops, n_envs, gamma=0.99, batch=8):
    np.random.seed(seed)
    rs = np.random.RandomState(seed + 5)
    rbuf = ReplayBuffer(capacity=capacity, num_steps=num_steps)
    obs_dim = 5
    obs_table = rs.randn(n_ops + 2, obs_dim).astype(np.float32)
    rec = dict(op_kind=[], op_a=[], op_b=[], reward=[], action=[], length=[])
    samples = dict(at_op=[], indices=[], entry_tids=[], entry_len=[], reward=[], terminal=[],
                   discount=[], action=[], state_tid=[], next_state_tid=[])
    tid = 0
    for k in range(n_ops):
        r = rs.rand()
        if r < 0.1:
            env = int(rs.randint(n_envs))
            rbuf.stop_current_episode(env_id=env)
            rec["op_kind"].append(1); rec["op_a"].append(env); rec["op_b"].append(0)
        else:
            env = int(rs.randint(n_envs))
            term = bool(rs.rand() < 0.12)
            rew = float(rs.randn())
            act = int(rs.randint(4))
            rbuf.append(state=obs_table[tid], action=act, reward=rew, next_state=obs_table[tid + 1],
                        is_state_terminal=term, env_id=env, tid=tid)
            rec["op_kind"].append(0); rec["op_a"].append(env); rec["op_b"].append(int(term))
            rec["reward"].append(rew); rec["action"].append(act)
            tid += 1
        rec["length"].append(len(rbuf))
        if len(rbuf) >= batch and rs.rand() < 0.2:
            st = np.random.get_state()
            exps = rbuf.sample(batch)
            np.random.set_state(st)
            idx = pfrl.utils.random.sample_n_k(len(rbuf), batch)
            samples["at_op"].append(k)
            samples["indices"].extend(int(i) for i in idx)
            be = batch_experiences(exps, torch.device("cpu"), lambda x: x, gamma)
            for e in exps:
                tids = [tr["tid"] for tr in e]
                samples["entry_len"].append(len(tids))
                samples["entry_tids"].extend(tids + [-1] * (num_steps - len(tids)))
                samples["state_tid"].append(e[0]["tid"])
                samples["next_state_tid"].append(e[-1]["tid"] + 1)
            samples["reward"].extend(be["reward"].numpy().tolist())
            samples["terminal"].extend(be["is_state_terminal"].numpy().tolist())
            samples["discount"].extend(be["discount"].numpy().tolist())
            samples["action"].extend(be["action"].numpy().tolist())
            # the batched states must be the rows of obs_table
            assert np.array_equal(be["state"].numpy(), obs_table[[e[0]["tid"] for e in exps]])
            assert np.array_equal(be["next_state"].numpy(),
                                  obs_table[[e[-1]["tid"] + 1 for e in exps]])
    final_entries = [[tr["tid"] for tr in e] for e in rbuf.memory]
    out = {k2: np.asarray(v) for k2, v in rec.items()}
    for k2, v in samples.items():
        dt = np.float32 if k2 in ("reward", "terminal", "discount") else None
        out["s_" + k2] = np.asarray(v, dtype=dt)
    out["final_len"] = np.array([len(e) for e in final_entries])
    out["final_tids"] = np.array(
        [e + [-1] * (num_steps - len(e)) for e in final_entries]).reshape(-1, num_steps)
    out["meta"] = np.array([seed, -1 if capacity is None else capacity, num_steps, n_envs, batch])
    out["gamma"] = np.array(gamma)
    np.savez_compressed(os.path.join(HERE, "replay_trace_%s.npz" % name), **out)
    print("replay_trace", name, "final len", len(rbuf), "samples", len(samples["at_op"]))


# --------------------------------------------------------------------------
# C. batch_states with the Atari phi
# --------------------------------------------------------------------------
def batch_states_golden():
    from pfrl.utils.batch_states import batch_states
    from pfrl.wrappers.atari_wrappers import LazyFrames

    def phi(x):  # examples/atari/train_dqn_batch_ale.py:229-231
        return np.asarray(x, dtype=np.float32) / 255

    lut = phi(np.arange(256, dtype=np.uint8))
    rs = np.random.RandomState(3)
    frames = rs.randint(0, 256, size=(7, 1, 12, 10)).astype(np.uint8)
    refs = np.array([[0, 0, 0, 0], [0, 0, 0, 1], [0, 1, 2, 3], [3, 4, 5, 6], [6, 6, 6, 6]],
                    dtype=np.int32)
    obs = [LazyFrames([frames[j] for j in r], stack_axis=0) for r in refs]
    out = batch_states(obs, torch.device("cpu"), phi).numpy()
    np.savez_compressed(os.path.join(HERE, "batch_states_atari.npz"), lut=lut, frames=frames,
                        refs=refs, out=out)
    print("batch_states", out.shape, out.dtype)


# --------------------------------------------------------------------------
# D2. GAE on the RECURRENT dataset: v_pred / next_v_pred stored as Python floats
# (ppo.py:98-107 `float(v)`), so every operation of ppo.py:36-47 is f64
# --------------------------------------------------------------------------
def gae_recurrent_golden():
    from pfrl.agents.ppo import _add_advantage_and_value_target_to_episode

    rs = np.random.RandomState(12)
    rec = dict(off=[0], reward=[], v=[], nv=[], nonterm=[], adv=[], vt=[], adv_t=[], gamma=[], lambd=[])
    for case in range(16):
        T = int(rs.randint(1, 140))
        gamma = [0.99, 1.0, 0.8, 0.0][case % 4]
        lambd = [0.95, 1.0, 0.0][case % 3]
        v = rs.randn(T).astype(np.float32)
        nv = rs.randn(T).astype(np.float32)
        rew = rs.choice([-1.0, 0.0, 1.0], size=T) if case % 3 else rs.randn(T)
        ep = []
        for i in range(T):
            done = (i == T - 1) and (case % 5 != 0)
            ep.append(dict(reward=(np.float64(rew[i]) if case % 2 else float(rew[i])),
                           nonterminal=0.0 if done else 1.0, v_pred=float(v[i]),
                           next_v_pred=float(nv[i])))
        _add_advantage_and_value_target_to_episode(ep, gamma, lambd)
        rec["off"].append(rec["off"][-1] + T)
        rec["reward"].extend(float(t["reward"]) for t in ep)
        rec["v"].extend(v.tolist()); rec["nv"].extend(nv.tolist())
        rec["nonterm"].extend(t["nonterminal"] for t in ep)
        rec["adv"].extend(float(t["adv"]) for t in ep)
        rec["vt"].extend(float(t["v_teacher"]) for t in ep)
        rec["adv_t"].extend(tag(t["adv"]) for t in ep)
        rec["gamma"].append(gamma); rec["lambd"].append(lambd)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["v"] = out["v"].astype(np.float32); out["nv"] = out["nv"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "gae_recurrent.npz"), **out)
    print("gae recurrent cases", len(rec["gamma"]))


# --------------------------------------------------------------------------
# D. GAE (ppo.py:36-47), both reward dtypes
# --------------------------------------------------------------------------
def gae_golden():
    from pfrl.agents.ppo import _add_advantage_and_value_target_to_episode

    rs = np.random.RandomState(11)
    rec = dict(off=[0], reward=[], v=[], nv=[], nonterm=[], adv=[], vt=[], adv_t=[], vt_t=[],
               mode=[], gamma=[], lambd=[])
    for case in range(24):
        T = int(rs.randint(1, 140))
        mode = case % 2
        gamma = [0.99, 1.0, 0.8, 0.0][case % 4]
        lambd = [0.95, 1.0, 0.0][case % 3]
        v = rs.randn(T).astype(np.float32)
        nv = rs.randn(T).astype(np.float32)
        rew = rs.choice([-1.0, 0.0, 1.0], size=T) if case % 3 else rs.randn(T)
        ep = []
        for i in range(T):
            done = (i == T - 1) and (case % 5 != 0)
            ep.append(dict(
                reward=(np.float64(rew[i]) if mode else float(rew[i])),
                nonterminal=0.0 if done else 1.0, v_pred=v[i], next_v_pred=nv[i]))
        _add_advantage_and_value_target_to_episode(ep, gamma, lambd)
        rec["off"].append(rec["off"][-1] + T)
        rec["reward"].extend(float(t["reward"]) for t in ep)
        rec["v"].extend(v.tolist()); rec["nv"].extend(nv.tolist())
        rec["nonterm"].extend(t["nonterminal"] for t in ep)
        rec["adv"].extend(float(t["adv"]) for t in ep)
        rec["vt"].extend(float(t["v_teacher"]) for t in ep)
        rec["adv_t"].extend(tag(t["adv"]) for t in ep)
        rec["vt_t"].extend(tag(t["v_teacher"]) for t in ep)
        rec["mode"].append(mode); rec["gamma"].append(gamma); rec["lambd"].append(lambd)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["v"] = out["v"].astype(np.float32); out["nv"] = out["nv"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "gae.npz"), **out)
    print("gae cases", len(rec["mode"]))


# --------------------------------------------------------------------------
# E. A2C._compute_returns (a2c.py:150-167)
# --------------------------------------------------------------------------
def a2c_golden():
    from pfrl.agents import A2C

    rs = np.random.RandomState(5)
    out = {}
    for ci, (T, N, use_gae, gamma, tau) in enumerate(
        [(5, 3, True, 0.99, 0.95), (5, 3, False, 0.99, 0.95), (16, 8, True, 0.9, 1.0),
         (16, 8, False, 1.0, 0.5)]):
        model = torch.nn.Linear(2, 2)
        agent = A2C(model, torch.optim.SGD(model.parameters(), lr=0.1), gamma=gamma,
                    num_processes=N, update_steps=T, use_gae=use_gae, tau=tau)
        agent.device = torch.device("cpu")
        agent._flush_storage((N, 2), torch.zeros(N, 1))
        agent.rewards = torch.tensor(rs.randn(T, N).astype(np.float32))
        agent.value_preds = torch.tensor(rs.randn(T + 1, N).astype(np.float32))
        agent.masks = torch.tensor((rs.rand(T, N) > 0.2).astype(np.float32))
        nvs = torch.tensor(rs.randn(N).astype(np.float32))
        vp_in = agent.value_preds.numpy().copy()
        agent._compute_returns(nvs)
        out["c%d_meta" % ci] = np.array([T, N, int(use_gae)])
        out["c%d_hyper" % ci] = np.array([gamma, tau])
        out["c%d_rewards" % ci] = agent.rewards.numpy()
        out["c%d_masks" % ci] = agent.masks.numpy()
        out["c%d_value_preds" % ci] = vp_in
        out["c%d_next_value" % ci] = nvs.numpy()
        out["c%d_returns" % ci] = agent.returns.numpy()
    np.savez_compressed(os.path.join(HERE, "a2c_returns.npz"), **out)
    print("a2c cases 4")


# --------------------------------------------------------------------------
# F. sample_n_k (utils/random.py) -- pins RNG stream use of our restatement
# --------------------------------------------------------------------------
def sample_n_k_golden():
    from pfrl.utils.random import sample_n_k

    rec = dict(n=[], k=[], seed=[], off=[0], idx=[])
    for seed, (n, k) in enumerate([(10, 3), (10, 4), (100, 32), (50000, 32), (1000000, 32),
                                   (33, 32), (5, 5), (7, 0), (97, 32), (96, 32), (40, 13)]):
        np.random.seed(seed)
        idx = sample_n_k(n, k)
        tail = np.random.random_sample()  # pins how much of the stream was consumed
        rec["n"].append(n); rec["k"].append(k); rec["seed"].append(seed)
        rec["idx"].extend(int(i) for i in idx)
        rec["off"].append(rec["off"][-1] + k)
        rec.setdefault("tail", []).append(tail)
    np.savez_compressed(os.path.join(HERE, "sample_n_k.npz"),
                        **{k: np.asarray(v) for k, v in rec.items()})
    print("sample_n_k cases", len(rec["n"]))


# --------------------------------------------------------------------------
# G. agent-level trace: reference DQN / DoubleDQN+PER driven by the reference's
#    train_agent_batch on a deterministic synthetic vector env.
# --------------------------------------------------------------------------
def make_q_function(n_in, n_actions, head):
    torch.manual_seed(1234)
    return torch.nn.Sequential(
        torch.nn.Flatten(), torch.nn.Linear(n_in, 32), torch.nn.ReLU(),
        torch.nn.Linear(32, n_actions), head)


def agent_trace(name, prioritized, num_steps, double, steps=640, N=4, agent_cls=None,
                **agent_kw):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv  # env only (numpy)

    import tempfile

    from pfrl import agents, explorers, experiments, replay_buffers
    from pfrl.q_functions import DiscreteActionValueHead

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=3, frame_shape=(12, 12), p_done=0.04)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    q = make_q_function(4 * 144, 6, DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2)
    if prioritized:
        rbuf = replay_buffers.PrioritizedReplayBuffer(
            200, alpha=0.5, beta0=0.4, betasteps=100, num_steps=num_steps,
            normalize_by_max="memory")
    else:
        rbuf = replay_buffers.ReplayBuffer(200, num_steps=num_steps)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    cls = agents.DoubleDQN if double else agents.DQN
    if agent_cls is not None:
        cls = getattr(agents, agent_cls, None) or getattr(agents.dpp, agent_cls)
    ag = cls(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40, minibatch_size=8,
             update_interval=4, target_update_interval=60, phi=phi, batch_accumulator="sum",
             **agent_kw)
    actions, losses, sampled = [], [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    ag.batch_act = spy_act
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        sampled.append([[float(np.asarray(t["reward"])) for t in e] for e in exps])
        orig_update(exps, errors_out)
        losses.append(ag.loss_record[-1])

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    flat_rewards = [sum(r for e in s for r in e) for s in sampled]
    out = dict(actions=np.asarray(actions), losses=np.asarray(losses),
               sampled_reward_sum=np.asarray(flat_rewards),
               sampled_len=np.asarray([[len(e) for e in s] for s in sampled]),
               final_params=np.concatenate([p.detach().numpy().ravel() for p in q.parameters()]),
               stats=np.asarray([float(v) for _, v in ag.get_statistics()]))
    if prioritized:
        out["final_tree_sum"] = np.asarray(float(rbuf.memory.priority_sums.sum()))
        out["final_max_priority"] = np.asarray(float(rbuf.memory.max_priority))
    np.savez_compressed(os.path.join(HERE, "agent_trace_%s.npz" % name), **out)
    print("agent_trace", name, "updates", len(losses), "final loss", losses[-1])


# --------------------------------------------------------------------------
# H. PPO trace: reference PPO on the synthetic vector env.  Actions are
#    recorded so that a run on another device (whose torch RNG differs) can
#    replay them; everything downstream (value pass, GAE, advantage
#    standardisation, minibatch order from `random`, losses) is then
#    comparable.
# --------------------------------------------------------------------------
# DQN variants that only change the target (advantage learning, dynamic policy programming)
DQN_FAMILY = [("al", "AL", dict(alpha=0.9)), ("pal", "PAL", dict(alpha=0.8)),
              ("double_pal", "DoublePAL", dict(alpha=0.9)), ("dpp", "DPP", dict(eta=2.0)),
              ("dppl", "DPPL", dict(eta=0.5)), ("dpp_greedy", "DPPGreedy", dict())]


def make_ppo_model(n_in, n_actions, SoftmaxCategoricalHead, Branched):
    torch.manual_seed(4321)
    return torch.nn.Sequential(
        torch.nn.Flatten(), torch.nn.Linear(n_in, 32), torch.nn.ReLU(),
        Branched(torch.nn.Sequential(torch.nn.Linear(32, n_actions), SoftmaxCategoricalHead()),
                 torch.nn.Linear(32, 1)))


def ppo_trace(name="ppo", steps=280, N=4):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, experiments
    from pfrl.policies import SoftmaxCategoricalHead

    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=5, frame_shape=(12, 12), p_done=0.06)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    model = make_ppo_model(4 * 144, 6, SoftmaxCategoricalHead, pfrl.nn.Branched)
    # plain SGD: Adam's normalised step amplifies CPU-vs-GPU fp32 rounding of
    # near-zero gradients to O(lr) parameter differences, which would hide the
    # data path under optimiser noise
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    ag = agents.PPO(model, opt, gpu=-1, gamma=0.99, lambd=0.95, phi=phi, update_interval=64,
                    minibatch_size=16, epochs=2, clip_eps=0.1, clip_eps_vf=None,
                    standardize_advantages=True, max_grad_norm=0.5)
    actions, losses, datasets = [], [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    ag.batch_act = spy_act
    orig_loss = ag._lossfun

    def spy_loss(*a, **kw):
        out = orig_loss(*a, **kw)
        losses.append([float(out), ag.value_loss_record[-1], ag.policy_loss_record[-1]])
        return out

    ag._lossfun = spy_loss
    orig_update = ag._update

    def spy_update(dataset):
        datasets.append(np.asarray([[float(b["adv"]), float(b["v_teacher"]), float(b["v_pred"]),
                                     float(b["log_prob"]), float(b["reward"]),
                                     float(b["nonterminal"])] for b in dataset]))
        return orig_update(dataset)

    ag._update = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    out = dict(actions=np.asarray(actions), losses=np.asarray(losses),
               final_params=np.concatenate([p.detach().numpy().ravel()
                                            for p in model.parameters()]),
               n_updates=np.asarray(ag.n_updates),
               explained_variance=np.asarray(ag.explained_variance))
    for i, d in enumerate(datasets):
        out["dataset%d" % i] = d
    out["n_datasets"] = np.asarray(len(datasets))
    np.savez_compressed(os.path.join(HERE, "agent_trace_%s.npz" % name), **out)
    print("ppo_trace updates", ag.n_updates, "datasets", len(datasets))


def _exp2(x):
    return torch.exp(2 * x)


def make_ppo_gaussian_model(obs_dim, act_dim, policies_mod, nn_mod):
    """The policy / value model of examples/mujoco/reproduction/ppo/train_ppo.py:112-134 in
    small: tanh MLP, state-independent diagonal covariance, separate value head."""
    torch.manual_seed(8642)
    return torch.nn.Sequential(
        torch.nn.Linear(obs_dim, 16), torch.nn.Tanh(),
        nn_mod.Branched(
            torch.nn.Sequential(
                torch.nn.Linear(16, act_dim),
                policies_mod.GaussianHeadWithStateIndependentCovariance(
                    action_size=act_dim, var_type="diagonal", var_func=_exp2, var_param_init=0)),
            torch.nn.Linear(16, 1)))


def ppo_mujoco_trace(steps=280, N=4, obs_dim=11, act_dim=3):
    """PPO with float32 vector observations, continuous actions and an
    EmpiricalNormalization obs_normalizer (clip 5), as the MuJoCo examples use it."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, experiments

    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=6, p_done=0.05)
    model = make_ppo_gaussian_model(obs_dim, act_dim, pfrl.policies, pfrl.nn)
    normalizer = pfrl.nn.EmpiricalNormalization(obs_dim, clip_threshold=5)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    ag = agents.PPO(model, opt, obs_normalizer=normalizer, gpu=-1, gamma=0.99, lambd=0.95,
                    update_interval=64, minibatch_size=16, epochs=2, clip_eps=0.2,
                    clip_eps_vf=None, standardize_advantages=True, entropy_coef=0.0,
                    max_grad_norm=0.5)
    actions, losses, datasets, norm_stats = [], [], [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append(np.asarray(a, dtype=np.float32))
        return a

    ag.batch_act = spy_act
    orig_loss = ag._lossfun

    def spy_loss(*a, **kw):
        out = orig_loss(*a, **kw)
        losses.append([float(out), ag.value_loss_record[-1], ag.policy_loss_record[-1]])
        return out

    ag._lossfun = spy_loss
    orig_update = ag._update

    def spy_update(dataset):
        datasets.append(np.asarray([[float(b["adv"]), float(b["v_teacher"]), float(b["v_pred"]),
                                     float(b["log_prob"])] for b in dataset]))
        r = orig_update(dataset)
        norm_stats.append(np.concatenate([normalizer.mean.numpy(), normalizer.std.numpy(),
                                          [float(normalizer.count)]]))
        return r

    ag._update = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    out = dict(actions=np.asarray(actions), losses=np.asarray(losses),
               final_params=np.concatenate([p.detach().numpy().ravel()
                                            for p in model.parameters()]),
               n_updates=np.asarray(ag.n_updates), norm_stats=np.asarray(norm_stats))
    for i, d in enumerate(datasets):
        out["dataset%d" % i] = d
    out["n_datasets"] = np.asarray(len(datasets))
    np.savez_compressed(os.path.join(HERE, "agent_trace_ppo_mujoco.npz"), **out)
    print("ppo_mujoco_trace updates", ag.n_updates, "datasets", len(datasets))


def a2c_trace(name="a2c", steps=120, N=4):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, experiments
    from pfrl.policies import SoftmaxCategoricalHead

    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    for use_gae in (True, False):
        pfrl.utils.set_random_seed(0)
        env = HostSyntheticAtariVectorEnv(N, seed=7, frame_shape=(12, 12), p_done=0.08)

        def phi(x):
            return np.asarray(x, dtype=np.float32) / 255

        model = make_ppo_model(4 * 144, 6, SoftmaxCategoricalHead, pfrl.nn.Branched)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        ag = agents.A2C(model, opt, gamma=0.99, num_processes=N, gpu=-1, update_steps=5, phi=phi,
                        use_gae=use_gae, tau=0.95, max_grad_norm=0.5)
        actions, returns = [], []
        orig_act = ag.batch_act

        def spy_act(obs, _orig=orig_act, _a=actions):
            a = _orig(obs)
            _a.append([int(x) for x in a])
            return a

        ag.batch_act = spy_act
        orig_upd = ag.update

        def spy_upd(_orig=orig_upd, _ag=ag, _r=returns):
            _orig()
            _r.append(_ag.returns.numpy().copy())

        ag.update = spy_upd
        experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
        out = dict(actions=np.asarray(actions), returns=np.asarray(returns),
                   final_params=np.concatenate([p.detach().numpy().ravel()
                                                for p in model.parameters()]),
                   stats=np.asarray([v for _, v in ag.get_statistics()]))
        np.savez_compressed(os.path.join(HERE, "agent_trace_%s_gae%d.npz" % (name, int(use_gae))),
                            **out)
        print("a2c_trace gae", use_gae, "updates", len(returns))


# --------------------------------------------------------------------------
# I. C51: categorical projection known answers + CategoricalDoubleDQN trace
# --------------------------------------------------------------------------
class DistNet(torch.nn.Module):
    """Tiny distributional Q-network: (B, 4, 12, 12) -> (B, 6, 11) probabilities."""

    def __init__(self, n_in=4 * 144, n_actions=6, n_atoms=11):
        super().__init__()
        torch.manual_seed(2468)
        self.l1 = torch.nn.Linear(n_in, 32)
        self.l2 = torch.nn.Linear(32, n_actions * n_atoms)
        self.n_actions, self.n_atoms = n_actions, n_atoms

    def forward(self, x):
        h = self.l2(torch.relu(self.l1(x.reshape(x.shape[0], -1))))
        return torch.softmax(h.reshape(-1, self.n_actions, self.n_atoms), dim=2)


def c51_projection_golden():
    from pfrl.agents.categorical_dqn import _apply_categorical_projection

    rs = np.random.RandomState(9)
    out = {}
    for ci, (B, n_atoms, vmin, vmax) in enumerate([(16, 51, -10.0, 10.0), (7, 11, -2.0, 2.0),
                                                   (4, 5, 0.0, 1.0)]):
        z = torch.linspace(vmin, vmax, n_atoms, dtype=torch.float32)
        y = torch.tensor(rs.uniform(vmin - 2, vmax + 2, size=(B, n_atoms)).astype(np.float32))
        y[0] = z                       # exactly on the support
        y[1, :] = vmax + 5.0           # all mass clipped to the top atom
        p = rs.rand(B, n_atoms).astype(np.float32)
        p /= p.sum(axis=1, keepdims=True)
        proj = _apply_categorical_projection(y, torch.tensor(p), z)
        out["c%d_z" % ci] = z.numpy()
        out["c%d_y" % ci] = y.numpy()
        out["c%d_p" % ci] = p
        out["c%d_proj" % ci] = proj.numpy()
    np.savez_compressed(os.path.join(HERE, "c51_projection.npz"), **out)
    print("c51 projection cases 3")


def _det_nets(obs_dim, act_dim, nn_mod, policies_mod):
    """Deterministic policy (tanh-bounded) and a Q-function for the DDPG / TD3 traces;
    built from the given package's modules so both sides construct identical nets."""
    policy = torch.nn.Sequential(
        torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(), torch.nn.Linear(32, act_dim),
        nn_mod.BoundByTanh(low=-np.ones(act_dim, dtype=np.float32),
                           high=np.ones(act_dim, dtype=np.float32)),
        policies_mod.DeterministicHead())

    def q():
        return torch.nn.Sequential(nn_mod.ConcatObsAndAction(),
                                   torch.nn.Linear(obs_dim + act_dim, 32), torch.nn.ReLU(),
                                   torch.nn.Linear(32, 1))

    return policy, q


def _shifted_smoothing(a):
    # deterministic stand-in for the clipped Gaussian target smoothing (the device and
    # host torch generators differ by construction)
    return torch.clamp(a + 0.05, -1, 1)


def td3_ddpg_traces(steps=260, N=2, obs_dim=24, act_dim=3):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, experiments, explorers, replay_buffers

    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    flat = lambda m: np.concatenate([p.detach().numpy().ravel() for p in m.parameters()])
    for kind in ("td3", "ddpg"):
        pfrl.utils.set_random_seed(0)
        env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=4, p_done=0.03)
        torch.manual_seed(2468)
        policy, q = _det_nets(obs_dim, act_dim, pfrl.nn, pfrl.policies)
        ex = explorers.AdditiveGaussian(scale=0.1, low=-1.0, high=1.0)
        burnin = lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32)
        rbuf = replay_buffers.ReplayBuffer(500)
        if kind == "td3":
            q1, q2 = q(), q()
            opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
            ag = agents.TD3(policy, q1, q2, opts[0], opts[1], opts[2], rbuf, gamma=0.99,
                            explorer=ex, gpu=-1, replay_start_size=40, minibatch_size=16,
                            update_interval=1, soft_update_tau=5e-3, burnin_action_func=burnin,
                            policy_update_delay=2,
                            target_policy_smoothing_func=_shifted_smoothing)
            crit, tgt = q1, ag.target_q_func1
            loss_of = lambda: [ag.q_func1_loss_record[-1], ag.q_func2_loss_record[-1]]
        else:
            q1 = q()
            opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1)]
            ag = agents.DDPG(policy, q1, opts[0], opts[1], rbuf, gamma=0.99, explorer=ex, gpu=-1,
                             replay_start_size=40, minibatch_size=16, update_interval=1,
                             target_update_interval=7, target_update_method="soft",
                             soft_update_tau=5e-2, burnin_action_func=burnin)
            crit, tgt = q1, ag.target_q_function
            loss_of = lambda: [ag.critic_loss_record[-1], ag.actor_loss_record[-1]]
        actions, losses = [], []
        orig_act = ag.batch_act

        def spy_act(obs, orig_act=orig_act):
            a = orig_act(obs)
            actions.append(np.asarray(a, dtype=np.float32))
            return a

        ag.batch_act = spy_act
        orig_update = ag.update

        def spy_update(exps, errors_out=None, orig_update=orig_update):
            orig_update(exps, errors_out)
            losses.append(loss_of())

        ag.replay_updater.update_func = spy_update
        experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
        np.savez_compressed(
            os.path.join(HERE, "agent_trace_%s.npz" % kind), actions=np.asarray(actions),
            losses=np.asarray(losses), policy_params=flat(policy), critic_params=flat(crit),
            target_critic_params=flat(tgt),
            stats=np.asarray([float(v) for _, v in ag.get_statistics()]))
        print(kind, "trace updates", len(losses))


def iqn_trace(steps=640, N=4, prioritized=True):
    """IQN + PrioritizedReplayBuffer(num_steps=3) on the CPU: thresholds come from the
    CPU torch generator (three draws per update + one per act), actions / losses /
    parameters recorded."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, explorers, experiments, replay_buffers
    from pfrl.agents import iqn as riqn

    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=13, frame_shape=(12, 12), p_done=0.04)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    torch.manual_seed(9753)
    q = riqn.ImplicitQuantileQFunction(
        psi=torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU()),
        phi=torch.nn.Sequential(riqn.CosineBasisLinear(16, 32), torch.nn.ReLU()),
        f=torch.nn.Linear(32, 6))
    opt = torch.optim.SGD(q.parameters(), lr=1e-2)
    if prioritized:
        rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                      num_steps=3, normalize_by_max="memory")
    else:
        rbuf = replay_buffers.ReplayBuffer(200)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    ag = agents.IQN(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40, minibatch_size=8,
                    update_interval=4, target_update_interval=60, phi=phi,
                    batch_accumulator="mean", quantile_thresholds_N=8,
                    quantile_thresholds_N_prime=8, quantile_thresholds_K=4)
    actions, losses = [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    ag.batch_act = spy_act
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        losses.append(float(ag.loss_record[-1]))

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    flat = np.concatenate([p.detach().numpy().ravel() for p in q.parameters()])
    name = "agent_trace_iqn_per_n3.npz" if prioritized else "agent_trace_iqn_uniform.npz"
    np.savez_compressed(os.path.join(HERE, name),
                        actions=np.asarray(actions), losses=np.asarray(losses),
                        final_params=flat,
                        stats=np.asarray([float(v) for _, v in ag.get_statistics()]))
    print("iqn trace updates", len(losses))


def cartpole_trace(steps=1500):
    """BASELINE configs[0]: examples/gym/train_dqn_gym.py settings (FC Q-function 100x2,
    Adam, ReplayBuffer(5e5), LinearDecayEpsilonGreedy) through the reference's
    train_agent on the CPU, on pfrl_amd's gym-free CartPole (numpy only)."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, explorers, experiments, q_functions, replay_buffers

    from pfrl_amd.envs.cartpole import CartPoleEnv

    pfrl.utils.set_random_seed(0)
    env = CartPoleEnv(seed=0)
    torch.manual_seed(77)
    q = q_functions.FCStateQFunctionWithDiscreteAction(4, 2, n_hidden_channels=100,
                                                       n_hidden_layers=2)
    opt = torch.optim.Adam(q.parameters())
    rbuf = replay_buffers.ReplayBuffer(5 * 10 ** 5)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 1000, env.action_space.sample)
    ag = agents.DQN(q, opt, rbuf, gpu=-1, gamma=0.99, explorer=ex, replay_start_size=200,
                    target_update_interval=100, update_interval=1, minibatch_size=32,
                    target_update_method="hard", soft_update_tau=1e-2)
    actions, rewards, losses = [], [], []
    orig_act = ag.act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append(int(a))
        return a

    ag.act = spy_act
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        losses.append(float(ag.loss_record[-1]))

    ag.replay_updater.update_func = spy_update
    experiments.train_agent(ag, env, steps, tempfile.mkdtemp(), max_episode_len=200)
    flat = np.concatenate([p.detach().numpy().ravel() for p in q.parameters()])
    np.savez_compressed(os.path.join(HERE, "agent_trace_cartpole_dqn.npz"),
                        actions=np.asarray(actions), losses=np.asarray(losses),
                        final_params=flat, final_state=env.state,
                        stats=np.asarray([float(v) for _, v in ag.get_statistics()]))
    print("cartpole trace updates", len(losses), "actions", len(actions))


def td_loss_golden():
    """(Double-)DQN target + Huber / MSE loss of the reference on fixed inputs
    (dqn.py:44-104, 424-470; double_dqn.py:15-40), with the gradient w.r.t. Q(s)."""
    from pfrl.action_value import DiscreteActionValue as DAV
    from pfrl.agents import dqn as rdqn

    out = {}
    ci = 0
    B, A = 16, 6
    for double in (False, True):
        for clip in (True, False):
            for acc in ("mean", "sum"):
                for weighted in (False, True):
                    g = torch.Generator().manual_seed(500 + ci)
                    q = (2 * torch.randn(B, A, generator=g)).requires_grad_(True)
                    tq = 2 * torch.randn(B, A, generator=g)
                    nq = torch.randn(B, A, generator=g)
                    act = torch.randint(0, A, (B,), generator=g)
                    r = torch.randn(B, generator=g)
                    disc = torch.full((B,), 0.99) ** torch.randint(1, 4, (B,), generator=g)
                    term = (torch.rand(B, generator=g) < 0.3).float()
                    w = torch.rand(B, generator=g) + 0.1
                    qout, tout = DAV(q), DAV(tq)
                    y = qout.evaluate_actions(act)
                    if double:
                        nxt = tout.evaluate_actions(DAV(nq).greedy_actions)
                    else:
                        nxt = tout.max
                    t = r + disc * (1.0 - term) * nxt
                    if weighted:
                        loss = rdqn.compute_weighted_value_loss(y, t, w, clip_delta=clip,
                                                                batch_accumulator=acc)
                    else:
                        loss = rdqn.compute_value_loss(y, t, clip_delta=clip,
                                                       batch_accumulator=acc)
                    (grad,) = torch.autograd.grad(loss, q)
                    pre = "k%d_" % ci
                    out.update({
                        pre + "q": q.detach().numpy(), pre + "tq": tq.numpy(), pre + "nq": nq.numpy(),
                        pre + "action": act.numpy(), pre + "reward": r.numpy(),
                        pre + "discount": disc.numpy(), pre + "terminal": term.numpy(),
                        pre + "weights": w.numpy(),
                        pre + "flags": np.asarray([int(double), int(clip), int(acc == "mean"),
                                                   int(weighted)]),
                        pre + "loss": np.asarray(loss.item(), dtype=np.float64),
                        pre + "grad": grad.numpy(), pre + "y": y.detach().numpy(),
                        pre + "t": t.numpy()})
                    ci += 1
    out["n_cases"] = np.asarray(ci)
    np.savez_compressed(os.path.join(HERE, "td_loss.npz"), **out)
    print("td loss cases", ci)


def c51_loss_golden():
    """Whole C51 loss path of the reference on fixed inputs: greedy next action,
    Bellman shift, projection, cross entropy, accumulation, gradient w.r.t. the
    online distribution (categorical_dqn.py:7-104,150-204; double: :10-52)."""
    from pfrl.action_value import DistributionalDiscreteActionValue as DAV
    from pfrl.agents import categorical_dqn as cd

    out = {}
    ci = 0
    combos = [(d, w, a) for d in (False, True)
              for w, a in ((False, "mean"), (False, "sum"), (True, "mean"), (True, "sum"))]
    big = [(False, False, "mean"), (True, True, "mean"), (True, False, "sum"), (False, True, "sum")]
    for (B, A, Z, vmin, vmax, sel) in [(16, 6, 51, -10.0, 10.0, big), (7, 3, 11, -2.0, 2.0, combos)]:
        for double, weighted, acc in sel:
            if True:
                g = torch.Generator().manual_seed(1000 * ci + 17)
                sm = lambda *s: torch.softmax(3 * torch.randn(*s, generator=g), dim=-1)
                q_dist = sm(B, A, Z).requires_grad_(True)
                next_dist, next_sel = sm(B, A, Z), sm(B, A, Z)
                z = torch.linspace(vmin, vmax, Z, dtype=torch.float32)
                action = torch.randint(0, A, (B,), generator=g)
                reward = torch.randint(-1, 2, (B,), generator=g).float() * 0.7
                discount = torch.full((B,), 0.99 ** 3)
                terminal = (torch.rand(B, generator=g) < 0.3).float()
                weights = torch.rand(B, generator=g) + 0.1
                qout, tq = DAV(q_dist, z), DAV(next_dist, z)
                greedy = (DAV(next_sel, z) if double else tq).greedy_actions
                Tz = (reward[..., None] + (1.0 - terminal[..., None]) * discount[..., None]
                      * z[None])
                t = cd._apply_categorical_projection(
                    Tz, tq.evaluate_actions_as_distribution(greedy).detach(), z)
                y = qout.evaluate_actions_as_distribution(action)
                elt = -t * torch.log(torch.clamp(y, 1e-10, 1.0))
                loss = (cd.compute_weighted_value_loss(elt, B, weights, acc) if weighted
                        else cd.compute_value_loss(elt, acc))
                (grad,) = torch.autograd.grad(loss, q_dist)
                pre = "k%d_" % ci
                out.update({
                    pre + "q_dist": q_dist.detach().numpy(), pre + "next_dist": next_dist.numpy(),
                    pre + "next_sel": next_sel.numpy(), pre + "z": z.numpy(),
                    pre + "action": action.numpy(), pre + "reward": reward.numpy(),
                    pre + "discount": discount.numpy(), pre + "terminal": terminal.numpy(),
                    pre + "weights": weights.numpy(),
                    pre + "flags": np.asarray([int(double), int(weighted), int(acc == "mean")]),
                    pre + "loss": np.asarray(loss.item(), dtype=np.float64),
                    pre + "grad": grad.numpy(),
                    pre + "delta": elt.detach().sum(dim=1).numpy(),
                    pre + "qsa": qout.evaluate_actions(action).detach().numpy(),
                    pre + "target": t.numpy()})
                ci += 1
    out["n_cases"] = np.asarray(ci)
    np.savez_compressed(os.path.join(HERE, "c51_loss.npz"), **out)
    print("c51 loss cases", ci)


def c51_agent_trace(steps=640, N=4):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, explorers, experiments, replay_buffers
    from pfrl.q_functions import DistributionalSingleModelStateQFunctionWithDiscreteAction

    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=11, frame_shape=(12, 12), p_done=0.04)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    q = DistributionalSingleModelStateQFunctionWithDiscreteAction(
        DistNet(), np.linspace(-3, 3, 11, dtype=np.float32))
    opt = torch.optim.SGD(q.parameters(), lr=1e-2)
    rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                  num_steps=3, normalize_by_max="memory")
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    ag = agents.CategoricalDoubleDQN(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40,
                                     minibatch_size=8, update_interval=4,
                                     target_update_interval=60, phi=phi, batch_accumulator="mean")
    actions, losses = [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    ag.batch_act = spy_act
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        losses.append(ag.loss_record[-1])

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    np.savez_compressed(
        os.path.join(HERE, "agent_trace_c51_per_n3.npz"), actions=np.asarray(actions),
        losses=np.asarray(losses),
        final_params=np.concatenate([p.detach().numpy().ravel() for p in q.parameters()]),
        final_tree_sum=np.asarray(float(rbuf.memory.priority_sums.sum())))
    print("c51 agent trace updates", len(losses))


# --------------------------------------------------------------------------
# J. SAC trace (vector observations, continuous actions).  The Gaussian noise
#    is switched off on both sides (sample == mean) because CPU and GPU torch
#    generators differ; everything else is the reference's update.
# --------------------------------------------------------------------------
def squashed_diagonal_gaussian_head(x):
    # examples/mujoco/reproduction/soft_actor_critic/train_soft_actor_critic.py:155-170
    from torch import distributions

    mean, log_scale = torch.chunk(x, 2, dim=1)
    log_scale = torch.clamp(log_scale, -20.0, 2.0)
    var = torch.exp(log_scale * 2)
    base = distributions.Independent(distributions.Normal(loc=mean, scale=torch.sqrt(var)), 1)
    return distributions.transformed_distribution.TransformedDistribution(
        base, [distributions.transforms.TanhTransform(cache_size=1)])


class _NoNoise:
    def __enter__(self):
        import torch.distributions as D

        self.saved = (D.Normal.rsample, D.Normal.sample)
        D.Normal.rsample = lambda self_, sample_shape=torch.Size(): self_.loc.expand(
            self_._extended_shape(sample_shape))
        D.Normal.sample = lambda self_, sample_shape=torch.Size(): self_.loc.expand(
            self_._extended_shape(sample_shape)).detach()

    def __exit__(self, *a):
        import torch.distributions as D

        D.Normal.rsample, D.Normal.sample = self.saved


def make_sac_nets(obs_dim, act_dim, ConcatObsAndAction, Lambda):
    torch.manual_seed(1357)
    policy = torch.nn.Sequential(torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(),
                                 torch.nn.Linear(32, act_dim * 2),
                                 Lambda(squashed_diagonal_gaussian_head))

    def q():
        return torch.nn.Sequential(ConcatObsAndAction(), torch.nn.Linear(obs_dim + act_dim, 32),
                                   torch.nn.ReLU(), torch.nn.Linear(32, 1))

    return policy, q(), q()


def sac_trace(steps=240, N=2, obs_dim=24, act_dim=3):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, experiments, replay_buffers

    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=2, p_done=0.03)
    policy, q1, q2 = make_sac_nets(obs_dim, act_dim, pfrl.nn.ConcatObsAndAction, pfrl.nn.Lambda)
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
    rbuf = replay_buffers.ReplayBuffer(500)
    ag = agents.SoftActorCritic(
        policy, q1, q2, opts[0], opts[1], opts[2], rbuf, gamma=0.99, gpu=-1,
        replay_start_size=40, minibatch_size=16, update_interval=1,
        burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
        entropy_target=None, initial_temperature=0.2, soft_update_tau=5e-3)
    actions, q1_losses = [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append(np.asarray(a, dtype=np.float32))
        return a

    ag.batch_act = spy_act
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        q1_losses.append([ag.q_func1_loss_record[-1], ag.q_func2_loss_record[-1]])

    ag.replay_updater.update_func = spy_update
    with _NoNoise():
        experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    flat = lambda m: np.concatenate([p.detach().numpy().ravel() for p in m.parameters()])
    np.savez_compressed(
        os.path.join(HERE, "agent_trace_sac.npz"), actions=np.asarray(actions),
        q_losses=np.asarray(q1_losses), policy_params=flat(policy), q1_params=flat(q1),
        target_q1_params=flat(ag.target_q_func1),
        stats=np.asarray([float(v) for _, v in ag.get_statistics()]))
    print("sac trace updates", len(q1_losses))



# --------------------------------------------------------------------------
# R. episodic replay, recurrent containers, persistent queues
# --------------------------------------------------------------------------
def episodic_trace(name, seed, capacity, n_ops, n_envs, batch=4, max_len=5):
    """Random append / stop traffic over several env_ids into the reference's
    EpisodicReplayBuffer; after each op the sizes, and at random points what
    sample_episodes (with and without max_len) and sample return, by transition id."""
    from pfrl.replay_buffers import EpisodicReplayBuffer

    np.random.seed(seed)
    rs = np.random.RandomState(seed + 5)
    rbuf = EpisodicReplayBuffer(capacity=capacity)
    rec = dict(op_kind=[], op_env=[], op_term=[], length=[], n_episodes=[])
    smp = dict(at_op=[], kind=[], ep_len=[], tids=[])
    tid = 0
    for k in range(n_ops):
        env = int(rs.randint(n_envs))
        if rs.rand() < 0.08:
            rbuf.stop_current_episode(env_id=env)
            rec["op_kind"].append(1); rec["op_env"].append(env); rec["op_term"].append(0)
        else:
            term = bool(rs.rand() < 0.15)
            rbuf.append(state=tid, action=tid % 3, reward=float(tid), next_state=tid + 1,
                        is_state_terminal=term, env_id=env, tid=tid)
            rec["op_kind"].append(0); rec["op_env"].append(env); rec["op_term"].append(int(term))
            tid += 1
        rec["length"].append(len(rbuf)); rec["n_episodes"].append(rbuf.n_episodes)
        if rbuf.n_episodes >= batch and rs.rand() < 0.25:
            which = int(rs.randint(3))
            if which == 0:
                eps = rbuf.sample_episodes(batch)
            elif which == 1:
                eps = rbuf.sample_episodes(batch, max_len=max_len)
            else:
                eps = rbuf.sample(batch)
            smp["at_op"].append(k); smp["kind"].append(which)
            for ep in eps:
                smp["ep_len"].append(len(ep))
                smp["tids"].extend(tr["tid"] for tr in ep)
    out = {k2: np.asarray(v, dtype=np.int64) for k2, v in rec.items()}
    out.update({"s_" + k2: np.asarray(v, dtype=np.int64) for k2, v in smp.items()})
    out["final_episode_len"] = np.asarray([len(ep) for ep in rbuf.episodic_memory], dtype=np.int64)
    out["final_tids"] = np.asarray([e[0]["tid"] for e in rbuf.memory], dtype=np.int64)
    out["meta"] = np.array([seed, -1 if capacity is None else capacity, n_envs, batch, max_len])
    np.savez_compressed(os.path.join(HERE, "episodic_trace_%s.npz" % name), **out)
    print("episodic_trace", name, "len", len(rbuf), "episodes", rbuf.n_episodes,
          "samples", len(smp["at_op"]))


def prioritized_episodic_trace(name, seed, capacity, n_ops, n_envs, batch=3, max_len=4,
                               normalize_by_max=True, uniform_ratio=0, prefix="prioritized_episodic_trace"):
    """The reference's PrioritizedEpisodicReplayBuffer under random traffic: sizes and
    capacity_left after every op; sampled (sub-)episodes, importance weights and the errors fed
    back, at random points."""
    from pfrl.replay_buffers import PrioritizedEpisodicReplayBuffer

    np.random.seed(seed)
    rs = np.random.RandomState(seed + 5)
    rbuf = PrioritizedEpisodicReplayBuffer(capacity=capacity, normalize_by_max=normalize_by_max,
                                           betasteps=50, error_max=2.0, uniform_ratio=uniform_ratio)
    rec = dict(op_kind=[], op_env=[], op_term=[], length=[], n_episodes=[], cap_left=[])
    smp = dict(at_op=[], ep_len=[], first_tid=[], weights=[], errors=[], beta=[])
    tid = 0
    for k in range(n_ops):
        env = int(rs.randint(n_envs))
        if rs.rand() < 0.08:
            rbuf.stop_current_episode(env_id=env)
            rec["op_kind"].append(1); rec["op_env"].append(env); rec["op_term"].append(0)
        else:
            term = bool(rs.rand() < 0.2)
            rbuf.append(state=tid, action=0, reward=0.0, next_state=tid + 1,
                        is_state_terminal=term, env_id=env, tid=tid)
            rec["op_kind"].append(0); rec["op_env"].append(env); rec["op_term"].append(int(term))
            tid += 1
        rec["length"].append(len(rbuf)); rec["n_episodes"].append(rbuf.n_episodes)
        rec["cap_left"].append(-1 if rbuf.capacity_left is None else rbuf.capacity_left)
        if rbuf.n_episodes >= batch and rs.rand() < 0.3:
            eps, weights = rbuf.sample_episodes(batch, max_len=max_len)
            errors = [float(e) for e in rs.rand(batch) * 3]
            rbuf.update_errors(errors)
            smp["at_op"].append(k)
            smp["ep_len"].extend(len(ep) for ep in eps)
            smp["first_tid"].extend(ep[0]["tid"] for ep in eps)
            smp["weights"].extend(float(w) for w in weights)
            smp["errors"].extend(errors)
            smp["beta"].append(rbuf.beta)
    out = {k2: np.asarray(v, dtype=np.int64) for k2, v in rec.items()}
    for k2, v in smp.items():
        out["s_" + k2] = np.asarray(v, dtype=np.float64 if k2 in ("weights", "errors", "beta")
                                    else np.int64)
    out["meta"] = np.array([seed, -1 if capacity is None else capacity, n_envs, batch, max_len])
    out["normalize"] = np.asarray({True: 1, "batch": 1, "memory": 2, False: 0}[normalize_by_max])
    out["uniform_ratio"] = np.asarray(float(uniform_ratio))
    np.savez_compressed(os.path.join(HERE, "%s_%s.npz" % (prefix, name)), **out)
    print(prefix, name, "len", len(rbuf), "episodes", rbuf.n_episodes,
          "samples", len(smp["at_op"]))


def make_recurrent_q_function(n_in, n_actions, nn_mod, head):
    torch.manual_seed(2468)
    tnn = torch.nn
    return nn_mod.RecurrentSequential(
        tnn.Flatten(), tnn.Linear(n_in, 32), tnn.ReLU(), tnn.LSTM(32, 16),
        tnn.Linear(16, n_actions), head)


def drqn_trace(steps=480, N=4):
    """examples/atari/train_drqn_ale.py in small: DoubleDQN(recurrent=True) over an
    EpisodicReplayBuffer, episodes cut to episodic_update_len, LSTM state stored per transition."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import tempfile

    from pfrl import agents, experiments, explorers, replay_buffers
    from pfrl.q_functions import DiscreteActionValueHead

    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=7, frame_shape=(12, 12), p_done=0.08)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    q = make_recurrent_q_function(4 * 144, 6, pfrl.nn, DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2)
    rbuf = replay_buffers.EpisodicReplayBuffer(300)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 300, lambda: np.random.randint(6))
    ag = agents.DoubleDQN(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40, minibatch_size=4,
                          update_interval=4, target_update_interval=60, phi=phi,
                          batch_accumulator="mean", recurrent=True, episodic_update_len=6)
    actions, losses, sampled = [], [], []
    orig_act = ag.batch_act

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    ag.batch_act = spy_act
    orig_update = ag.update_from_episodes

    def spy_update(episodes, errors_out=None):
        sampled.append([len(ep) for ep in episodes])
        orig_update(episodes, errors_out)
        losses.append(ag.loss_record[-1])

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    with ag.eval_mode():                       # evaluation keeps its own recurrent state
        obs = env.reset()
        eval_actions = []
        for _ in range(6):
            a = orig_act(obs)
            obs, r, done, info = env.step(a)
            ag.batch_observe(obs, r, done, [False] * N)
            eval_actions.append([int(x) for x in a])
    out = dict(actions=np.asarray(actions), losses=np.asarray(losses),
               sampled_len=np.asarray(sampled), eval_actions=np.asarray(eval_actions),
               final_params=np.concatenate([p.detach().numpy().ravel() for p in q.parameters()]),
               stats=np.asarray([float(v) for _, v in ag.get_statistics()]),
               rlen=np.asarray([len(rbuf), rbuf.n_episodes]))
    np.savez_compressed(os.path.join(HERE, "agent_trace_drqn.npz"), **out)
    print("drqn_trace updates", len(losses), "final loss", losses[-1], "episodes", rbuf.n_episodes)


def rmsprop_eps_inside_sqrt_golden():
    """Four steps of the reference's RMSpropEpsInsideSqrt on two tensors, for the plain,
    centered, momentum and weight-decay configurations."""
    import warnings

    rs = np.random.RandomState(17)
    w0 = [rs.randn(3, 4).astype(np.float32), rs.randn(5).astype(np.float32)]
    grads = [[rs.randn(*w.shape).astype(np.float32) for w in w0] for _ in range(4)]
    out = {"w0_0": w0[0], "w0_1": w0[1]}
    for t, gs in enumerate(grads):
        out["g%d_0" % t], out["g%d_1" % t] = gs
    configs = dict(plain=dict(), centered=dict(centered=True), momentum=dict(momentum=0.9),
                   decay=dict(weight_decay=0.01, centered=True, momentum=0.5))
    for name, kw in configs.items():
        ps = [torch.nn.Parameter(torch.from_numpy(w.copy())) for w in w0]
        cls = (pfrl.optimizers.SharedRMSpropEpsInsideSqrt if name == "momentum"
               else pfrl.optimizers.RMSpropEpsInsideSqrt)
        opt = cls(ps, lr=7e-4, eps=1e-1, alpha=0.99, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for t, gs in enumerate(grads):
                for p_, g_ in zip(ps, gs):
                    p_.grad = torch.from_numpy(g_.copy())
                opt.step()
                for i, p_ in enumerate(ps):
                    out["%s_w%d_%d" % (name, t, i)] = p_.detach().numpy().copy()
        for i, p_ in enumerate(ps):
            for key, v in opt.state[p_].items():
                out["%s_state_%s_%d" % (name, key, i)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "rmsprop_eps_inside_sqrt.npz"), **out)
    print("rmsprop_eps_inside_sqrt golden:", len(out), "arrays")


def atari_wrappers_golden(steps=400):
    """The reference's wrapper stack (minus WarpFrame: cv2 is absent) over the scripted FakeALE:
    per step the stacked observation's checksum and newest frame, reward, done, needs_reset,
    and how often the game itself was reset / stepped."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from _fake_ale import FakeALE

    from pfrl.wrappers import ContinuingTimeLimit, atari_wrappers as aw

    out = {}
    for name, fire, flicker, scale in (("plain", False, False, False),
                                       ("fire_flicker_scaled", True, True, True)):
        game = FakeALE(seed=3)
        env = ContinuingTimeLimit(game, max_episode_steps=90)
        env = aw.MaxAndSkipEnv(aw.NoopResetEnv(env, noop_max=5), skip=4)
        env = aw.EpisodicLifeEnv(env)
        if fire:
            env = aw.FireResetEnv(env)
        if scale:
            env = aw.ScaledFloatFrame(env)
        env = aw.ClipRewardEnv(env)
        if flicker:
            env = aw.FlickerFrame(env)
        env = aw.FrameStack(env, 4, channel_order="chw")
        rs = np.random.RandomState(11)
        rec = dict(obs_sum=[], newest=[], reward=[], done=[], needs_reset=[], resets=[], steps=[],
                   shared=[])
        obs = env.reset()
        prev = obs
        for t in range(steps):
            a = int(rs.randint(3))
            obs, r, done, info = env.step(a)
            rec["obs_sum"].append(float(np.asarray(obs, dtype=np.float64).sum()))
            rec["newest"].append(np.asarray(obs)[-1].astype(np.float32))
            rec["reward"].append(float(r)); rec["done"].append(bool(done))
            rec["needs_reset"].append(bool(info.get("needs_reset", False)))
            rec["shared"].append(sum(a_ is b_ for a_, b_ in zip(obs._frames[:-1], prev._frames[1:])))
            if done or info.get("needs_reset", False):
                obs = env.reset()
            prev = obs
            rec["resets"].append(game.n_resets); rec["steps"].append(game.n_steps)
        for k, v in rec.items():
            out["%s_%s" % (name, k)] = np.asarray(v)
        out["%s_space_low" % name] = env.observation_space.low
        out["%s_space_high" % name] = env.observation_space.high
    np.savez_compressed(os.path.join(HERE, "atari_wrappers.npz"), **out)
    print("atari_wrappers golden: dones", int(out["plain_done"].sum()),
          "game resets", int(out["plain_resets"][-1]))


def api_signatures_golden():
    """Parameter names, kinds and literal defaults of every callable of the boundary."""
    import json

    sys.path.insert(0, os.path.join(HERE, ".."))
    from _api_surface import describe_api

    desc = describe_api("pfrl")
    with open(os.path.join(HERE, "api_signatures.json"), "w") as f:
        json.dump(desc, f, indent=0, sort_keys=True)
    print("api signatures:", len(desc), "callables")


def episodic_golden():
    episodic_trace("unbounded", 30, None, 300, 3)
    episodic_trace("cap40", 31, 40, 600, 4)
    episodic_trace("cap7", 32, 7, 400, 2, batch=2, max_len=2)
    prioritized_episodic_trace("cap30", 40, 30, 500, 3)
    prioritized_episodic_uniform_traces()
    prioritized_episodic_trace("unbounded_memory", 41, None, 300, 2, normalize_by_max="memory")
    prioritized_episodic_trace("cap12_nonorm", 42, 12, 300, 2, batch=2, normalize_by_max=False)


def make_recurrent_model(nn_mod):
    """LSTM trunk between stateless layers, then a branched head (one branch recurrent)."""
    tnn = torch.nn
    return nn_mod.RecurrentSequential(
        tnn.Linear(5, 8), tnn.ReLU(), tnn.LSTM(8, 6),
        nn_mod.RecurrentBranched(
            tnn.GRU(6, 4),
            nn_mod.RecurrentSequential(tnn.Linear(6, 3), tnn.Tanh())))


def recurrent_golden():
    """RecurrentSequential / RecurrentBranched outputs and states for packed sequences and for a
    one-step batch continuing them, the masking / indexing / stacking helpers, and
    batch_recurrent_experiences on episodes that carry stored recurrent states."""
    from pfrl.replay_buffer import batch_recurrent_experiences
    from pfrl.utils import recurrent as R

    torch.manual_seed(123)
    rs = np.random.RandomState(9)
    model = make_recurrent_model(pfrl.nn)
    out = {"sd_" + k: v.numpy() for k, v in model.state_dict().items()}
    lens = [4, 2, 2, 1]
    seqs = [torch.from_numpy(rs.randn(n, 5).astype(np.float32)) for n in lens]
    with torch.no_grad():
        (y_gru, y_mlp), state = R.pack_and_forward(model, seqs, None)
        step_in = torch.from_numpy(rs.randn(len(lens), 5).astype(np.float32))
        masked = R.mask_recurrent_state_at(state, [1, 3])
        (z_gru, z_mlp), state2 = R.one_step_forward(model, step_in, masked)
    picked = R.get_recurrent_state_at(state2, 2, detach=True)
    restacked = R.concatenate_recurrent_states(
        [R.get_recurrent_state_at(state2, i, detach=True) if i != 1 else None for i in range(4)])

    def flat(prefix, tree):
        leaves = []

        def walk(t):
            if isinstance(t, tuple):
                for u in t:
                    walk(u)
            else:
                leaves.append(t)
        walk(tree)
        for i, leaf in enumerate(leaves):
            out["%s_%d" % (prefix, i)] = leaf.numpy()

    out["lens"] = np.asarray(lens)
    for i, s in enumerate(seqs):
        out["seq_%d" % i] = s.numpy()
    out["step_in"] = step_in.numpy()
    out["y_gru"], out["y_mlp"] = y_gru.numpy(), y_mlp.numpy()
    out["z_gru"], out["z_mlp"] = z_gru.numpy(), z_mlp.numpy()
    flat("state", state); flat("masked", masked); flat("state2", state2)
    flat("picked", picked); flat("restacked", restacked)

    # batch_recurrent_experiences: three episodes, sorted by length, LSTM-shaped stored states
    def rstate():
        return (rs.randn(1, 6).astype(np.float32), rs.randn(1, 6).astype(np.float32))

    episodes, tid = [], 0
    for n in (3, 2, 1):
        ep = []
        for j in range(n):
            ep.append(dict(state=rs.randn(5).astype(np.float32), action=int(rs.randint(4)),
                           reward=float(rs.randn()), next_state=rs.randn(5).astype(np.float32),
                           next_action=int(rs.randint(4)), is_state_terminal=(j == n - 1 and n != 2),
                           recurrent_state=rstate() if (j or n != 1) else None,
                           next_recurrent_state=rstate()))
            tid += 1
        episodes.append(ep)
    be = batch_recurrent_experiences(episodes, torch.device("cpu"), lambda x: x, 0.97)
    for key in ("action", "reward", "is_state_terminal", "discount", "next_action"):
        out["be_" + key] = be[key].numpy()
    for i in range(3):
        out["be_state_%d" % i] = be["state"][i].numpy()
        out["be_next_state_%d" % i] = be["next_state"][i].numpy()
    flat("be_rs", be["recurrent_state"]); flat("be_nrs", be["next_recurrent_state"])
    out["ep_lens"] = np.asarray([len(ep) for ep in episodes])
    col = lambda key, dt: np.asarray([tr[key] for ep in episodes for tr in ep], dtype=dt)  # noqa: E731
    out["ep_state"], out["ep_next_state"] = col("state", np.float32), col("next_state", np.float32)
    out["ep_action"], out["ep_next_action"] = col("action", np.int64), col("next_action", np.int64)
    out["ep_reward"], out["ep_terminal"] = col("reward", np.float64), col("is_state_terminal", np.int64)
    for i, ep in enumerate(episodes):
        for key in ("recurrent_state", "next_recurrent_state"):
            s0 = ep[0][key]
            out["ep%d_%s_none" % (i, key)] = np.asarray(s0 is None)
            if s0 is not None:
                out["ep%d_%s_h" % (i, key)], out["ep%d_%s_c" % (i, key)] = s0
    np.savez_compressed(os.path.join(HERE, "recurrent.npz"), **out)
    print("recurrent golden:", len(out), "arrays")


PERSISTENT_ITEMS = [[dict(state=np.arange(3, dtype=np.float32) + i, action=i % 2, reward=0.5 * i,
                          next_state=np.arange(3, dtype=np.float32) + i + 1, next_action=None,
                          is_state_terminal=(i % 4 == 3))] for i in range(11)]


def persistent_golden():
    """Directories written by the reference's PersistentRandomAccessQueue: ``base`` in two
    sessions (5 + 3 items) with small chunks so that generations rotate, and ``child`` whose
    ancestor is ``base`` (3 more items).  Paths inside meta.pkl are relative to tests/golden and
    the timestamp is pinned, so regeneration is byte-stable."""
    import datetime as _dt
    import shutil

    import pfrl.collections.persistent_collections as pc

    class PinnedClock:
        @staticmethod
        def today():
            return _dt.datetime(2024, 8, 7, 12, 0, 0)

        @staticmethod
        def strftime(d, fmt):
            return d.strftime(fmt)

    pc.datetime = PinnedClock

    class SmallChunks(pc.PersistentRandomAccessQueue):
        chunk_size = 700      # bytes; an item pickles to ~330 B => a generation holds 3 items

    root = os.path.join(HERE, "persistent_queue")
    shutil.rmtree(root, ignore_errors=True)
    cwd = os.getcwd()
    os.chdir(HERE)
    try:
        q = SmallChunks("persistent_queue/base", 6)
        for item in PERSISTENT_ITEMS[:5]:
            q.append(item)
        q.close()
        q = SmallChunks("persistent_queue/base", 6)
        assert len(q) == 5
        q.extend(PERSISTENT_ITEMS[5:8])
        assert len(q) == 6
        q.close()
        c = SmallChunks("persistent_queue/child", 4, ancestor="persistent_queue/base")
        assert len(c) == 4
        for item in PERSISTENT_ITEMS[8:]:
            c.append(item)
        c.close()
    finally:
        os.chdir(cwd)
    print("persistent golden:", sorted(os.listdir(os.path.join(root, "base", "rank0"))),
          sorted(os.listdir(os.path.join(root, "child", "rank0"))))


if __name__ == "__main__":
    if sys.argv[1:]:          # regenerate selected fixtures only: make_golden.py recurrent_golden ...
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    random.seed(0)
    torch.manual_seed(0)
    pbuf_trace("cap5", 0, 5, 400, 2)
    pbuf_trace("cap1", 1, 1, 120, 1)
    pbuf_trace("cap10", 2, 10, 500, 4)
    pbuf_trace("cap64", 3, 64, 1500, 16)
    pbuf_trace("cap1000", 4, 1000, 3000, 32)
    pbuf_trace("unbounded", 5, None, 700, 8)
    pbuf_uniform_traces()
    per_trace("f32_cap100_n1", 10, 100, 1, 600, 8, "f32", True)
    per_trace("f32_cap50_n3", 11, 50, 3, 600, 8, "f32", "memory", alpha=0.5)
    per_trace("py_cap20_n1", 12, 20, 1, 300, 4, "py", False)
    per_trace("mixed_cap300_n3", 13, 300, 3, 1500, 32, "mixed", "memory", alpha=0.5, n_envs=8)
    replay_trace("cap30_n1", 20, 30, 1, 400, 3)
    replay_trace("cap30_n3", 21, 30, 3, 500, 4)
    replay_trace("unbounded_n5", 22, None, 5, 300, 2)
    replay_trace("cap200_n3_env16", 23, 200, 3, 1200, 16, batch=32)
    batch_states_golden()
    gae_golden()
    gae_recurrent_golden()
    a2c_golden()
    sample_n_k_golden()
    agent_trace("dqn_uniform_n1", False, 1, False)
    agent_trace("ddqn_per_n3", True, 3, True)
    ppo_trace()
    ppo_mujoco_trace()
    a2c_trace()
    cartpole_trace()
    for fam in DQN_FAMILY:
        agent_trace(fam[0], False, 1, False, agent_cls=fam[1], **fam[2])
    iqn_trace(prioritized=True)
    iqn_trace(prioritized=False)
    td3_ddpg_traces()
    c51_projection_golden()
    td_loss_golden()
    c51_loss_golden()
    c51_agent_trace()
    sac_trace()
    episodic_golden()
    recurrent_golden()
    persistent_golden()
    drqn_trace()
    rmsprop_eps_inside_sqrt_golden()
    atari_wrappers_golden()
    api_signatures_golden()
