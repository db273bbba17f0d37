"""ctypes front-end of the CPU oracle (oracle/pfrl_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing
under ``pfrl_amd/`` imports this package; the product path fails loudly when
its HIP library is missing instead of falling back to this code.

Parity status: pinned.  ``tests/test_oracle_golden.py`` checks every entry
point against vectors recorded from the reference itself by
``tests/golden/make_golden.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

T_NONE, T_PY, T_F32, T_F64 = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpfrl_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "pfrl_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        P = c.c_void_p
        L.orc_pbuf_create.restype = P
        L.orc_pbuf_create.argtypes = [c.c_long]
        L.orc_pbuf_destroy.argtypes = [P]
        L.orc_pbuf_len.restype = c.c_long
        L.orc_pbuf_len.argtypes = [P]
        L.orc_pbuf_popleft.restype = c.c_int64
        L.orc_pbuf_popleft.argtypes = [P]
        L.orc_pbuf_append.argtypes = [P, c.c_int64, c.c_double, c.c_int]
        L.orc_pbuf_sample.argtypes = [P, c.c_long] + [P] * 9
        L.orc_pbuf_set_last_priority.argtypes = [P, c.c_long, P, P]
        L.orc_pbuf_stats.argtypes = [P, P, P, P]
        L.orc_pbuf_dump_level.restype = c.c_long
        L.orc_pbuf_dump_level.argtypes = [P, c.c_int, c.c_long, P, P]
        L.orc_nstep_create.restype = P
        L.orc_nstep_create.argtypes = [c.c_int, c.c_int]
        L.orc_nstep_destroy.argtypes = [P]
        L.orc_nstep_append.restype = c.c_long
        L.orc_nstep_append.argtypes = [P, c.c_int, c.c_long, c.c_int]
        L.orc_nstep_stop.restype = c.c_long
        L.orc_nstep_stop.argtypes = [P, c.c_int]
        L.orc_nstep_emitted.argtypes = [P, P, P]
        L.orc_batch_experiences_scalars.argtypes = [c.c_long, c.c_int] + [P] * 10
        L.orc_batch_states_u8.argtypes = [c.c_long, c.c_int, c.c_long, P, P, c.c_float, P]
        L.orc_batch_states_f32.argtypes = [c.c_long, c.c_int, c.c_long, P, P, P]
        L.orc_gae_fragment.argtypes = [c.c_long, P, P, P, P, c.c_double, c.c_double, c.c_int, P, P]
        L.orc_a2c_returns.argtypes = [c.c_long, c.c_long, P, P, P, P, c.c_double, c.c_double, c.c_int]
        L.orc_priority_from_errors_f32.argtypes = [
            c.c_long, P, c.c_int, c.c_double, c.c_int, c.c_double, c.c_double, c.c_double, P, P,
        ]
        L.orc_adv_stats.argtypes = [c.c_long, P, P, P]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def type_tag(x):
    """NEP-50 type tag of a Python/NumPy scalar as the reference would see it."""
    if x is None:
        return T_NONE
    if isinstance(x, np.float32):
        return T_F32
    if isinstance(x, np.float64):
        return T_F64
    if isinstance(x, (float, int)):
        return T_PY
    raise TypeError(type(x))


class OraclePrioritizedBuffer:
    """pfrl.collections.prioritized.PrioritizedBuffer restated (uniform_ratio=0)."""

    def __init__(self, capacity=None):
        self._h = lib().orc_pbuf_create(-1 if capacity is None else int(capacity))
        self.capacity = capacity

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pbuf_destroy(self._h)
            self._h = None

    def __len__(self):
        return lib().orc_pbuf_len(self._h)

    def append(self, payload, priority=None):
        lib().orc_pbuf_append(
            self._h, int(payload), 0.0 if priority is None else float(priority), type_tag(priority)
        )

    def popleft(self):
        return lib().orc_pbuf_popleft(self._h)

    def sample(self, u01):
        u01 = np.ascontiguousarray(u01, dtype=np.float64)
        n = len(u01)
        idx = np.zeros(n, dtype=np.int64)
        payload = np.zeros(n, dtype=np.int64)
        pri = np.zeros(n, dtype=np.float64)
        ptag = np.zeros(n, dtype=np.int32)
        prob = np.zeros(n, dtype=np.float64)
        total = np.zeros(1, dtype=np.float64)
        ttag = np.zeros(1, dtype=np.int32)
        minp = np.zeros(1, dtype=np.float64)
        lib().orc_pbuf_sample(
            self._h, n, _p(u01), _p(idx), _p(payload), _p(pri), _p(ptag), _p(prob), _p(total),
            _p(ttag), _p(minp),
        )
        return dict(
            indices=idx, payload=payload, priorities=pri, priority_tags=ptag, probabilities=prob,
            total=float(total[0]), total_tag=int(ttag[0]), min_prob=float(minp[0]),
        )

    def set_last_priority(self, values, tags):
        values = np.ascontiguousarray(values, dtype=np.float64)
        tags = np.ascontiguousarray(tags, dtype=np.int32)
        lib().orc_pbuf_set_last_priority(self._h, len(values), _p(values), _p(tags))

    def stats(self):
        v = np.zeros(3, dtype=np.float64)
        t = np.zeros(3, dtype=np.int32)
        i = np.zeros(3, dtype=np.int64)
        lib().orc_pbuf_stats(self._h, _p(v), _p(t), _p(i))
        return dict(
            sum=(v[0], int(t[0])), min=(v[1], int(t[1])), max_priority=(v[2], int(t[2])),
            length=int(i[0]), bounds=(int(i[1]), int(i[2])),
        )

    def dump_level(self, which, width):
        st = self.stats()
        size = st["bounds"][1] - st["bounds"][0]
        n = max(size // width, 0)
        v = np.zeros(max(n, 1), dtype=np.float64)
        t = np.zeros(max(n, 1), dtype=np.int32)
        got = lib().orc_pbuf_dump_level(self._h, which, width, _p(v), _p(t))
        return v[:got], t[:got]


class OracleNStep:
    """Per-env n-step windows of pfrl.replay_buffers.ReplayBuffer (ids only)."""

    def __init__(self, num_steps, max_envs=1024):
        self.n = num_steps
        self._h = lib().orc_nstep_create(num_steps, max_envs)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_nstep_destroy(self._h)
            self._h = None

    def _emitted(self, k):
        tids = np.zeros((max(k, 1), self.n), dtype=np.int64)
        lens = np.zeros(max(k, 1), dtype=np.int32)
        lib().orc_nstep_emitted(self._h, _p(tids), _p(lens))
        return [list(tids[i, : lens[i]]) for i in range(k)]

    def append(self, env, tid, terminal):
        k = lib().orc_nstep_append(self._h, int(env), int(tid), int(bool(terminal)))
        return self._emitted(k)

    def stop(self, env):
        k = lib().orc_nstep_stop(self._h, int(env))
        return self._emitted(k)


def batch_experiences_scalars(entries, rewards, terminals, gamma, n):
    """entries: list of lists of transition ids."""
    B = len(entries)
    tids = -np.ones((B, n), dtype=np.int64)
    lens = np.zeros(B, dtype=np.int32)
    for b, e in enumerate(entries):
        tids[b, : len(e)] = e
        lens[b] = len(e)
    rewards = np.ascontiguousarray(rewards, dtype=np.float64)
    terminals = np.ascontiguousarray(terminals, dtype=np.uint8)
    gp = np.array([gamma**i for i in range(n + 1)], dtype=np.float64)
    out_r = np.zeros(B, dtype=np.float32)
    out_t = np.zeros(B, dtype=np.float32)
    out_d = np.zeros(B, dtype=np.float32)
    first = np.zeros(B, dtype=np.int64)
    last = np.zeros(B, dtype=np.int64)
    lib().orc_batch_experiences_scalars(
        B, n, _p(tids), _p(lens), _p(rewards), _p(terminals), _p(gp), _p(out_r), _p(out_t),
        _p(out_d), _p(first), _p(last),
    )
    return dict(reward=out_r, is_state_terminal=out_t, discount=out_d, first=first, last=last)


def batch_states_u8(frames, refs, scale_div=255.0):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    refs = np.ascontiguousarray(refs, dtype=np.int32)
    M, k = refs.shape
    fe = int(np.prod(frames.shape[1:]))
    out = np.empty((M, k, fe), dtype=np.float32)
    lib().orc_batch_states_u8(M, k, fe, _p(frames), _p(refs), float(scale_div), _p(out))
    return out


def batch_states_f32(frames, refs):
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    refs = np.ascontiguousarray(refs, dtype=np.int32)
    M, k = refs.shape
    fe = int(np.prod(frames.shape[1:]))
    out = np.empty((M, k, fe), dtype=np.float32)
    lib().orc_batch_states_f32(M, k, fe, _p(frames), _p(refs), _p(out))
    return out


def gae_fragment(reward, v_pred, next_v_pred, nonterminal, gamma, lambd, mode=0):
    T = len(reward)
    reward = np.ascontiguousarray(reward, dtype=np.float64)
    v_pred = np.ascontiguousarray(v_pred, dtype=np.float32)
    next_v_pred = np.ascontiguousarray(next_v_pred, dtype=np.float32)
    nonterminal = np.ascontiguousarray(nonterminal, dtype=np.float64)
    adv = np.zeros(T, dtype=np.float64)
    vt = np.zeros(T, dtype=np.float64)
    lib().orc_gae_fragment(
        T, _p(reward), _p(v_pred), _p(next_v_pred), _p(nonterminal), float(gamma), float(lambd),
        int(mode), _p(adv), _p(vt),
    )
    return adv, vt


def a2c_returns(rewards, masks, value_preds, next_value, gamma, tau, use_gae):
    T, N = rewards.shape
    rewards = np.ascontiguousarray(rewards, dtype=np.float32)
    masks = np.ascontiguousarray(masks, dtype=np.float32)
    vp = np.ascontiguousarray(value_preds, dtype=np.float32).copy()
    ret = np.zeros((T + 1, N), dtype=np.float32)
    if use_gae:
        vp[T] = next_value
    else:
        ret[T] = next_value
    lib().orc_a2c_returns(T, N, _p(rewards), _p(masks), _p(vp), _p(ret), float(gamma), float(tau),
                          int(use_gae))
    return ret


def priority_from_errors_f32(err, error_min, error_max, eps, alpha):
    err = np.ascontiguousarray(err, dtype=np.float32)
    n = len(err)
    v = np.zeros(n, dtype=np.float64)
    t = np.zeros(n, dtype=np.int32)
    lib().orc_priority_from_errors_f32(
        n, _p(err), int(error_min is not None), float(error_min or 0), int(error_max is not None),
        float(error_max or 0), float(eps), float(alpha), _p(v), _p(t),
    )
    return v, t


def adv_stats(adv):
    adv = np.ascontiguousarray(adv, dtype=np.float32)
    m = np.zeros(1)
    s = np.zeros(1)
    lib().orc_adv_stats(len(adv), _p(adv), _p(m), _p(s))
    return float(m[0]), float(s[0])


def c51_loss(q_dist, action, next_dist, next_select, z, reward, discount, terminal, weights, mean):
    """Whole C51 loss path (orc_c51_loss): returns dict(loss, grad, qsa, delta, target)."""
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    q_dist, next_dist, next_select = f32(q_dist), f32(next_dist), f32(next_select)
    z, reward, discount, terminal, weights = (f32(a) for a in (z, reward, discount, terminal,
                                                              weights))
    action = np.ascontiguousarray(action, dtype=np.int64)
    B, A, Z = q_dist.shape
    loss = ctypes.c_double()
    grad = np.empty((B, A, Z), dtype=np.float32)
    qsa = np.empty(B, dtype=np.float32)
    delta = np.empty(B, dtype=np.float32)
    target = np.empty((B, Z), dtype=np.float32)
    L = lib()
    L.orc_c51_loss.restype = None
    L.orc_c51_loss.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_long] * 3 + [ctypes.c_int] + \
        [ctypes.c_void_p] * 5
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    L.orc_c51_loss(ptr(q_dist), ptr(action), ptr(next_dist), ptr(next_select), ptr(z), ptr(reward),
                   ptr(discount), ptr(terminal), ptr(weights), B, A, Z, int(bool(mean)),
                   ctypes.byref(loss), ptr(grad), ptr(qsa), ptr(delta), ptr(target))
    return dict(loss=loss.value, grad=grad, qsa=qsa, delta=delta, target=target)


def dqn_td_loss(q, action, target_q, next_q_online, reward, discount, terminal, weights,
                clip_delta, mean):
    """(Double-)DQN TD loss (orc_dqn_td_loss): dict(loss, grad, y, t)."""
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    q, target_q, next_q_online = f32(q), f32(target_q), f32(next_q_online)
    reward, discount, terminal, weights = (f32(a) for a in (reward, discount, terminal, weights))
    action = np.ascontiguousarray(action, dtype=np.int64)
    B, A = q.shape
    loss = ctypes.c_double()
    grad = np.empty((B, A), dtype=np.float32)
    y = np.empty(B, dtype=np.float32)
    t = np.empty(B, dtype=np.float32)
    L = lib()
    L.orc_dqn_td_loss.restype = None
    L.orc_dqn_td_loss.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_long] * 2 + \
        [ctypes.c_int] * 2 + [ctypes.c_void_p] * 4
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    L.orc_dqn_td_loss(ptr(q), ptr(action), ptr(target_q), ptr(next_q_online), ptr(reward),
                      ptr(discount), ptr(terminal), ptr(weights), B, A, int(bool(clip_delta)),
                      int(bool(mean)), ctypes.byref(loss), ptr(grad), ptr(y), ptr(t))
    return dict(loss=loss.value, grad=grad, y=y, t=t)
