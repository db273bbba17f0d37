"""The native host step planner (csrc/hostplan.hip) against the Python walk it replaces: same
values AND same position of NumPy's global stream afterwards (everything downstream shares
that stream).  Reference: pfrl/explorers/epsilon_greedy.py:8-12, pfrl/utils/random.py:4-28,
pfrl/agents/dqn.py:516-549, pfrl/replay_buffers/replay_buffer.py:33-62."""
import collections

import numpy as np
import pytest

from pfrl_amd import _native, host_plan
from pfrl_amd.utils.random import sample_n_k as py_sample_n_k


def _state_eq(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def _reference_sample_n_k(n, k):
    """pfrl/utils/random.py:4-28 verbatim in behaviour (set walk)."""
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


def test_native_bounded_draw_equals_numpy_randint():
    """randint(n) for n from 1 to beyond 2^32: values and stream position."""
    for n in [1, 2, 3, 6, 7, 8, 18, 255, 256, 257, 10 ** 6, 2 ** 31, 2 ** 32 - 1, 2 ** 32,
              2 ** 32 + 1, 2 ** 40 + 12345]:
        # epsilon = 2: rand() < 2 always fires -> eps_greedy makes rand(), randint(n) per env;
        # the bounded draw alone is what sample_n_k's first pass consumes
        if n <= 0x7fffffff:
            np.random.seed(n % 1000)
            want2 = []
            for _ in range(64):
                np.random.rand()
                want2.append(np.random.randint(n))
            end2 = np.random.get_state()
            np.random.seed(n % 1000)
            got = host_plan.eps_greedy(64, 2.0, n)
            assert np.array_equal(got, np.asarray(want2)), n
            assert _state_eq(np.random.get_state(), end2), n
        if n > 3 * 16:
            np.random.seed(n % 1000)
            first = np.random.randint(0, n, size=32)
            np.random.seed(n % 1000)
            got = host_plan.sample_n_k(n, 16)
            if len(set(first[:16].tolist())) == 16:
                assert np.array_equal(got, first[:16]), n


def test_native_eps_greedy_equals_the_python_loop():
    from pfrl_amd import explorers

    for seed, eps, n_act in [(0, 0.3, 6), (1, 1.0, 18), (2, 0.0, 4), (3, 0.01, 6), (4, 0.5, 1)]:
        ex = explorers.ConstantEpsilonGreedy(eps, lambda: np.random.randint(n_act))
        np.random.seed(seed)
        want = [ex.select_action(0, lambda: -1) for _ in range(1000)]
        end = np.random.get_state()
        np.random.seed(seed)
        got = host_plan.eps_greedy(1000, eps, n_act)
        assert got.tolist() == [int(w) for w in want]
        assert _state_eq(np.random.get_state(), end)


def test_native_sample_n_k_differential_1e5_calls():
    """10^5 calls against the reference's set walk, sizes from the duplicate-heavy edge
    (n just above 3 k: repairs and spare refills on most calls) to replay-sized."""
    rs = np.random.RandomState(7)
    np.random.seed(11)
    calls = 0
    repaired = 0
    for it in range(2000):
        k = int(rs.choice([1, 2, 5, 8, 32, 64]))
        n = int(rs.choice([3 * k + 1, 3 * k + 2, 4 * k, 10 * k, 1000 * k + 17, 10 ** 6]))
        st = np.random.get_state()
        want = [_reference_sample_n_k(n, k) for _ in range(25)]
        end = np.random.get_state()
        np.random.set_state(st)
        got = [host_plan.sample_n_k(n, k) for _ in range(25)]
        for w, g in zip(want, got):
            assert np.array_equal(w, g), (n, k)
            assert len(set(g.tolist())) == k
        assert _state_eq(np.random.get_state(), end), (n, k)
        np.random.set_state(st)
        mine = [py_sample_n_k(n, k) for _ in range(25)]       # the package's vectorised form
        for w, g in zip(want, mine):
            assert np.array_equal(w, g), (n, k)
        assert _state_eq(np.random.get_state(), end), (n, k)
        calls += 50
        np.random.set_state(st)
        repaired += int(any(len(set(np.random.randint(0, n, size=2 * k)[:k].tolist())) < k
                            for _ in range(25)))
        np.random.set_state(end)
    assert calls == 10 ** 5 and repaired > 300


def test_recognise_randint():
    n_actions = 6
    assert host_plan.recognise_randint(lambda: np.random.randint(n_actions)) == 6
    assert host_plan.recognise_randint(lambda: np.random.randint(18)) == 18
    assert host_plan.recognise_randint(lambda: np.random.randint(0, 4)) == 4
    rs = np.random.RandomState(0)
    assert host_plan.recognise_randint(lambda: rs.randint(6)) is None          # own generator
    calls = []

    class Sampler:                                 # e.g. gym's action_space.sample: never called
        def sample(self):
            calls.append(1)
            return 0

    assert host_plan.recognise_randint(Sampler().sample) is None and not calls
    assert host_plan.recognise_randint(lambda: int(np.random.rand() * 6)) is None
    assert host_plan.recognise_randint(lambda: np.random.uniform(-1, 1, size=2)) is None
    st = np.random.get_state()
    host_plan.recognise_randint(lambda: np.random.randint(9))
    assert _state_eq(np.random.get_state(), st)                                # stream untouched


class _FakeFrames:
    def __init__(self):
        self.live = 0

    def oldest_live_seq(self):
        return self.live


class _FakeStore:
    """The host mirrors of DeviceReplayStore, without a device."""

    def __init__(self, R, E, k, bound):
        self.R, self.E, self.k, self.n, self.bound = R, E, k, 1, bound
        self.act_dim, self.desc = 0, object()
        self.h_state_ref = np.zeros((R, k), np.int32)
        self.h_next_ref = np.zeros((R, k), np.int32)
        self.h_reward = np.zeros(R, np.float64)
        self.h_terminal = np.zeros(R, np.uint8)
        self.h_min_fseq = np.zeros(R, np.int64)
        self.h_e_tids = -np.ones((E, 1), np.int64)
        self.h_e_len = np.zeros(E, np.int32)
        self.h_e_min_fseq = np.zeros(E, np.int64)
        self.n_trans = self.n_entries = 0
        self.frames = _FakeFrames()


class _FakeQueue:
    def __init__(self, maxlen):
        self.maxlen, self.head = maxlen, 0


class _FakeBuffer:
    def __init__(self, store, maxlen):
        self.store, self.memory = store, _FakeQueue(maxlen)


@pytest.mark.parametrize("capacity", [300, None])
def test_plan_dqn_range_equals_the_reference_loop(capacity):
    """256-env steps cut into ranges: appended rows, host mirrors, queue head, every index set
    (as entry ring slots) and the stream position equal a deque + sample_n_k restatement of
    pfrl/agents/dqn.py:516-549 with ReplayBuffer(capacity, num_steps=1)."""
    R = E = 420
    k, B, N = 4, 8, 64
    store = _FakeStore(R, E, k, bound=400)
    rbuf = _FakeBuffer(store, capacity)
    plan = host_plan.DQNRangePlanner(rbuf)
    block = np.zeros(1 << 16, dtype=np.uint8)
    rs = np.random.RandomState(3)
    ref_q = collections.deque(maxlen=capacity)     # holds entry seqs
    np.random.seed(5)
    t = 0
    seq = 0
    fseq = 0
    replay_start, interval = 40, 4
    updates = 0
    for step in range(9 if capacity else 5):
        s_refs = rs.randint(0, 1000, size=(N, k)).astype(np.int32)
        n_refs = rs.randint(0, 1000, size=(N, k)).astype(np.int32)
        s_min = np.arange(fseq, fseq + N, dtype=np.int64)
        n_min = s_min + rs.randint(-2, 3, size=N)
        fseq += N
        reward = rs.randn(N)
        done = (rs.rand(N) < 0.1).astype(np.uint8)
        for lo, hi in ((0, 6), (6, 25), (25, N)):
            m = hi - lo
            # expectation (reference order): append, then the draws of a due update
            st0 = np.random.get_state()
            want_sets, t_run = [], t
            for j in range(lo, hi):
                t_run += 1
                ref_q.append(seq)
                seq += 1
                if len(ref_q) >= replay_start and t_run % interval == 0:
                    idx = _reference_sample_n_k(len(ref_q), B)
                    want_sets.append([ref_q[int(i)] % E for i in idx])
            end = np.random.get_state()
            np.random.set_state(st0)
            U = plan.plan(np.ascontiguousarray(s_refs[lo:hi]), np.ascontiguousarray(s_min[lo:hi]),
                          np.ascontiguousarray(n_refs[lo:hi]), np.ascontiguousarray(n_min[lo:hi]),
                          np.ascontiguousarray(reward[lo:hi]), np.ascontiguousarray(done[lo:hi]),
                          t, replay_start, interval, 1, B, block)
            t = t_run
            assert U == len(want_sets)
            assert _state_eq(np.random.get_state(), end)
            o = plan.offs
            got = block[o[8]:o[8] + 4 * U * B].view(np.int32).reshape(U, B)
            assert got.tolist() == want_sets
            updates += U
            tsl = block[o[0]:o[0] + 4 * m].view(np.int32)
            tid0 = store.n_trans - m
            assert tsl.tolist() == [(tid0 + j) % R for j in range(m)]
            assert np.array_equal(block[o[1]:o[1] + 16 * m].view(np.int32).reshape(m, k), s_refs[lo:hi])
            assert np.array_equal(block[o[2]:o[2] + 16 * m].view(np.int32).reshape(m, k), n_refs[lo:hi])
            assert np.array_equal(block[o[3]:o[3] + 8 * m].view(np.float64), reward[lo:hi])
            assert np.array_equal(block[o[4]:o[4] + m], done[lo:hi])
            assert block[o[5]:o[5] + 4 * m].view(np.int32).tolist() == \
                [(store.n_entries - m + j) % E for j in range(m)]
            assert np.array_equal(block[o[6]:o[6] + 4 * m].view(np.int32), tsl)
            assert (block[o[7]:o[7] + 4 * m].view(np.int32) == 1).all()
            assert np.array_equal(store.h_state_ref[tsl], s_refs[lo:hi])
            assert np.array_equal(store.h_min_fseq[tsl], np.minimum(s_min, n_min)[lo:hi])
            assert store.n_entries == seq and rbuf.memory.head == ref_q[0]
            assert store.n_entries - rbuf.memory.head == len(ref_q)
    assert updates > 50


def test_plan_dqn_range_leaves_everything_alone_in_the_dense_regime():
    store = _FakeStore(64, 64, 4, 64)
    rbuf = _FakeBuffer(store, 50)
    plan = host_plan.DQNRangePlanner(rbuf)
    block = np.zeros(1 << 14, dtype=np.uint8)
    z = lambda *s, d=np.int32: np.zeros(s, dtype=d)   # noqa: E731
    np.random.seed(0)
    st = np.random.get_state()
    rc = plan.plan(z(8, 4), z(8, d=np.int64), z(8, 4), z(8, d=np.int64), z(8, d=np.float64),
                   z(8, d=np.uint8), 0, 4, 4, 1, 8, block)      # 3 * 8 >= len
    assert rc == host_plan.PLAN_DENSE
    assert _state_eq(np.random.get_state(), st)
    assert store.n_trans == 0 and store.n_entries == 0 and rbuf.memory.head == 0


def test_synth_reward_done_native_equals_numpy():
    from pfrl_amd.envs.synthetic import reward_done_stream

    lib = _native.lib()
    for seed, id0, n, t in [(0, 0, 256, 1), (3, 512, 100, 123456), (2 ** 40, 7, 33, 2 ** 33)]:
        r = np.empty(n)
        d = np.empty(n, dtype=np.uint8)
        _native.check(lib.pfrl_synth_reward_done(seed, id0, n, t, 1.0 / 500, r.ctypes.data,
                                                 d.ctypes.data))
        wr, wd = reward_done_stream(seed, np.arange(id0, id0 + n), t, 1.0 / 500)
        assert np.array_equal(r, wr) and np.array_equal(d.astype(bool), wd)
