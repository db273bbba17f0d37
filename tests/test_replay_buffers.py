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
