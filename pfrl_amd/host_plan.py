"""Python face of the native HOST step planner (csrc/hostplan.hip, include/pfrl_amd.h).

The reference's batched DQN step consumes NumPy's legacy global stream in a fixed order
(/root/reference/pfrl/agents/dqn.py:490-549): per env ``rand()`` [+ ``randint(n_actions)``] in
``batch_act`` (pfrl/explorers/epsilon_greedy.py:8-12), then per env ``append`` and, when an
update is due, ``sample_n_k(len, B)`` (pfrl/replay_buffer.py:329-356, pfrl/utils/random.py:4-28).
At 256 envs that walk is ~1.5 ms of interpreter time per step.  The planner makes the same draws
on NumPy's own generator object through the C function table NumPy publishes (``bitgen_t``), so
the stream position before and after is exactly that of the Python loop.
"""
import ctypes

import numpy as np

from pfrl_amd import _native

PLAN_DENSE, PLAN_OVERFLOW, PLAN_FRAME_RING = -10, -11, -12


def bitgen_ptr():
    """``bitgen_t *`` of the generator behind ``np.random.*`` (the legacy global RandomState).
    Looked up on every call: ``np.random.seed`` / ``set_state`` act on this object in place."""
    return np.random.mtrand._rand._bit_generator.ctypes.bit_generator.value


def sample_n_k(n, k):
    """``pfrl.utils.random.sample_n_k`` in the sparse regime (3 k < n); None otherwise."""
    if not (1 <= k <= 4096 and 3 * k < n):
        return None
    out = np.empty(k, dtype=np.int64)
    _native.check(_native.lib().pfrl_plan_sample_n_k(bitgen_ptr(), int(n), int(k), out.ctypes.data),
                  "plan_sample_n_k")
    return out


def eps_greedy(n_envs, epsilon, n_actions, out=None):
    """n_envs x ``select_action_epsilon_greedily`` with ``random_action_func = lambda:
    np.random.randint(n_actions)``: int32 [n_envs], the random action or -1 = greedy."""
    if out is None:
        out = np.empty(n_envs, dtype=np.int32)
    _native.check(_native.lib().pfrl_plan_eps_greedy(bitgen_ptr(), int(n_envs), float(epsilon),
                                                     int(n_actions), out.ctypes.data),
                  "plan_eps_greedy")
    return out


_randint_cache = {}


def recognise_randint(func, probes=48, max_actions=1 << 16):
    """n such that ``func()`` behaves as ``np.random.randint(n)`` on the global stream (the
    ``random_action_func`` of the example scripts: examples/atari/train_dqn_batch_ale.py:214-218
    ``lambda: np.random.randint(n_actions)``), or None (e.g. ``action_space.sample``, which
    has its own generator).  Decided by behaviour, like ``recognise_phi``: ``func`` is called
    ``probes`` times from a known stream state; for a candidate n, ``np.random.randint(0, n,
    probes)`` from the same state (element for element the scalar draw) must give the same
    values AND leave the stream at the same position.  The global stream is restored
    afterwards.  The planner's native draw is in turn pinned to ``np.random.randint`` by
    tests/test_host_plan.py."""
    key = id(func)
    hit = _randint_cache.get(key)
    if hit is not None and hit[0] is func:
        return hit[1]
    # Only a plain Python function that names ``randint`` is probed at all: calling an unknown
    # callable 48 times could advance a generator of its own (``action_space.sample``) and change
    # the run.  Anything else keeps the Python loop.
    code = getattr(func, "__code__", None)
    if code is None or "randint" not in code.co_names:
        _randint_cache[key] = (func, None)
        return None
    n_found = None
    saved = np.random.get_state()
    try:
        np.random.seed(0x5EED)
        start = np.random.get_state()
        try:
            vals = [func() for _ in range(probes)]
        except Exception:
            vals = None
        end = np.random.get_state()
        ok = vals is not None and all(
            isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)) for v in vals)
        if ok and min(vals) >= 0 and max(vals) < max_actions:
            want = np.asarray(vals, dtype=np.int64)
            for n in range(max(vals) + 1, max(vals) + 66):
                np.random.set_state(start)
                got = np.random.randint(0, n, size=probes)
                st = np.random.get_state()
                if np.array_equal(got, want) and st[2] == end[2] and np.array_equal(st[1], end[1]):
                    n_found = n
                    break
    finally:
        np.random.set_state(saved)
    if len(_randint_cache) > 64:
        _randint_cache.clear()
    _randint_cache[key] = (func, n_found)
    return n_found


class DQNRangePlanner:
    """Binds the planner to one uniform, one-step device ReplayBuffer (store mirrors, queue
    head) and plans env ranges of ``DQN._batch_observe_train`` -- and of the same per-env loop
    of the vector-observation agents (pfrl/agents/soft_actor_critic.py:354-374, td3.py:283-303,
    ddpg.py:207-227: append, then ``sample_n_k`` when an update is due) -- into a pinned staging
    block."""

    def __init__(self, rbuf, float_actions=False):
        st = rbuf.store
        # the planner never touches the action column: DQN's comes from the device tensor of
        # selected actions, the vector-observation agents' (float_actions) rides in the same
        # pinned block behind the planner's part (agents/_vector_device_step.py)
        assert st.n == 1 and st.desc is not None and (st.act_dim == 0 or float_actions)
        self.rbuf = rbuf
        self.store = st
        d = _native.HostStoreDesc()
        for name in ("h_state_ref", "h_next_ref", "h_reward", "h_terminal", "h_min_fseq",
                     "h_e_tids", "h_e_len", "h_e_min_fseq"):
            arr = getattr(st, name)
            assert arr.flags["C_CONTIGUOUS"]
            setattr(d, name, arr.ctypes.data)
        assert st.h_e_tids.dtype == np.int64 and st.h_e_len.dtype == np.int32
        d.R, d.E = st.R, st.E
        d.maxlen = -1 if rbuf.memory.maxlen is None else int(rbuf.memory.maxlen)
        d.bound = st.bound
        d.k, d.n = st.k, st.n
        self.desc = d
        self._mirrors = tuple(getattr(st, n) for n in ("h_state_ref", "h_e_tids"))
        self.counters = np.zeros(3, dtype=np.int64)
        self.offs = np.zeros(10, dtype=np.int64)
        self._fn = _native.lib().pfrl_plan_dqn_range

    def valid_for(self, rbuf):
        st = rbuf.store
        return (rbuf is self.rbuf and st is self.store and st.h_state_ref is self._mirrors[0]
                and st.h_e_tids is self._mirrors[1])

    def block_bytes(self, m, U, B):
        k = self.store.k
        return 16 * 10 + 4 * m * (4 + 2 * k) + 8 * m + m + 4 * U * B

    def plan(self, s_refs, s_min_seq, n_refs, n_min_seq, reward, done, t0, replay_start,
             update_interval, n_times_update, B, host_block):
        """Returns U (>= 0) with ``self.offs`` filled, or PLAN_DENSE (nothing touched)."""
        st, q = self.store, self.rbuf.memory
        m = len(reward)
        c = self.counters
        c[0], c[1], c[2] = st.n_trans, st.n_entries, q.head
        oldest = st.frames.oldest_live_seq()
        rc = self._fn(ctypes.byref(self.desc), bitgen_ptr(), m, s_refs.ctypes.data,
                      s_min_seq.ctypes.data, n_refs.ctypes.data, n_min_seq.ctypes.data,
                      reward.ctypes.data, done.ctypes.data, int(t0), int(replay_start),
                      int(update_interval), int(n_times_update), int(B), int(oldest),
                      c.ctypes.data, host_block.ctypes.data, host_block.nbytes,
                      self.offs.ctypes.data)
        if rc == PLAN_DENSE:
            return PLAN_DENSE
        st.n_trans, st.n_entries, q.head = int(c[0]), int(c[1]), int(c[2])
        if rc < 0:
            raise RuntimeError("pfrl_amd plan_dqn_range failed (code %d): %s"
                               % (rc, _native.lib().pfrl_amd_last_error().decode()))
        return int(rc)
