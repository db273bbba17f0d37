"""Device-resident observation storage.

The reference keeps observations as Python objects: ``VectorFrameStack`` hands
out ``LazyFrames`` that share the last k-1 frame arrays by identity
(/root/reference/pfrl/wrappers/vector_frame_stack.py:93-105,
atari_wrappers.py:251-272) and every consumer re-materialises them with
``np.concatenate``.  Here a frame is written ONCE into a ring in HBM and an
observation is just k slot numbers (int32); consecutive observations share
slots instead of array identities.
"""
import numpy as np
import torch

from pfrl_amd import ops


class DeviceFrameStore:
    """Ring of ``n_slots`` frames of ``frame_shape`` / ``dtype`` in HBM."""

    def __init__(self, n_slots, frame_shape, dtype=torch.uint8, device=None, stack=4):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceFrameStore lives in HBM; got device %s" % device)
        assert dtype in (torch.uint8, torch.float32)
        self.n_slots = int(n_slots)
        self.frame_shape = tuple(int(s) for s in frame_shape)
        self.dtype = dtype
        self.stack = int(stack)
        nbytes = int(np.prod(self.frame_shape)) * (1 if dtype == torch.uint8 else 4)
        if nbytes % 4:
            raise ValueError("frame size must be a multiple of 4 bytes, got %d" % nbytes)
        self.frame_bytes = nbytes
        self.frames = torch.zeros((self.n_slots,) + self.frame_shape, dtype=dtype,
                                  device=self.device)
        self.next_seq = 0  # frames written so far; slot = seq % n_slots

    def ref_staging(self, nbytes):
        """Pinned staging ring for observation-ref uploads (a few KB per acting step)."""
        ring = getattr(self, "_ref_ring", None)
        if ring is None or ring.slot_bytes < nbytes + 64:
            from pfrl_amd.staging import StagingRing

            ring = self._ref_ring = StagingRing(self.device, slot_bytes=max(1 << 14, 2 * nbytes + 64),
                                                n_slots=16)
        return ring

    def alloc(self, n):
        """Reserve ``n`` consecutive ring positions -> (seqs int64, slots int32)."""
        seqs = np.arange(self.next_seq, self.next_seq + n, dtype=np.int64)
        self.next_seq += n
        return seqs, (seqs % self.n_slots).astype(np.int32)

    def oldest_live_seq(self):
        return self.next_seq - self.n_slots

    def write(self, src, slots_dev):
        """frames[slots] <- src (device tensor [n, *frame_shape])."""
        ops.frames_scatter(self.frames, src.contiguous(), slots_dev)

    # Set by an agent whose network runs in torch.channels_last: stacks of four u8
    # planes are then gathered straight into that memory format (pfrl_*_nhwc4) and the
    # network's per-pass NCHW -> NHWC conversion of the minibatch disappears.
    emit_channels_last = False

    def gather(self, refs_dev, divisor=255.0, out=None):
        if self.emit_channels_last and ops.channels_last_supported(self.frames, refs_dev.shape[1]):
            return ops.batch_states_nhwc4(self.frames, refs_dev, divisor, out=out)
        x = ops.batch_states(self.frames, refs_dev, divisor, out=out)
        # the observation as the host would see it: LazyFrames concatenates its (1, H, W) frames
        # on axis 0 (pfrl/wrappers/atari_wrappers.py:262-266), a single frame is the observation
        k, fs = refs_dev.shape[1], self.frame_shape
        if k == 1:
            return x.view((x.shape[0],) + fs)
        if len(fs) >= 2 and fs[0] == 1:
            return x.view((x.shape[0], k) + fs[1:])
        return x


class DeviceObs:
    """One observation = k frame slots of a DeviceFrameStore."""

    __slots__ = ("store", "refs", "min_seq")

    def __init__(self, store, refs, min_seq):
        self.store = store
        self.refs = refs        # numpy int32 [k]
        self.min_seq = min_seq  # oldest frame sequence number in the stack

    def to_numpy(self):
        """Materialise on the host (API compatibility / debugging; one D2H)."""
        idx = torch.from_numpy(self.refs.astype(np.int64)).to(self.store.device)
        x = self.store.frames[idx].cpu().numpy()
        if x.shape[0] == 1:
            return x[0]
        # LazyFrames concatenates on axis 0 (atari_wrappers.py:262-266)
        if x.ndim >= 3 and x.shape[1] == 1:
            return np.concatenate(list(x), axis=0)
        return x

    def __array__(self, dtype=None, copy=None):
        out = self.to_numpy()
        return out.astype(dtype) if dtype is not None else out


class DeviceObsBatch:
    """A batch of observations living in a DeviceFrameStore; what a device
    VectorEnv returns from reset()/step().  Behaves as a sequence of
    DeviceObs so drivers written for lists of observations keep working."""

    __slots__ = ("store", "refs", "min_seq", "_refs_dev")

    def __init__(self, store, refs, min_seq, refs_dev=None):
        self.store = store
        self.refs = refs          # numpy int32 [N, k]
        self.min_seq = min_seq    # numpy int64 [N]
        self._refs_dev = refs_dev

    def __len__(self):
        return self.refs.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return DeviceObsBatch(self.store, self.refs[i], self.min_seq[i])
        return DeviceObs(self.store, self.refs[i], self.min_seq[i])

    def __iter__(self):
        for i in range(len(self)):
            yield DeviceObs(self.store, self.refs[i], self.min_seq[i])

    def refs_device(self, staging=None):
        if self._refs_dev is None:
            if staging is None:
                # pinned, asynchronous: a pageable .to(device) is a blocking copy (~0.25 ms
                # of host time per acting step at 256 envs)
                staging = self.store.ref_staging(self.refs.nbytes)
            (self._refs_dev,) = staging.upload([self.refs])
        return self._refs_dev


class DeviceActions:
    """The actions of one batched step, resident on the device (int64 [N]).

    What ``DQN.batch_act`` returns on the device fast path: epsilon-greedy is resolved on the
    GPU from the host's draws (``ops.select_actions``), the replay append reads the action
    column straight from this tensor, and nothing waits for a D2H copy.  A driver or env that
    looks at the values (``env.step(actions)`` of a host env: indexing, iteration,
    ``np.asarray``) gets them through ONE copy made on first use -- the reference's
    ``batch_argmax = ....detach().cpu().numpy()`` (pfrl/agents/dqn.py:497), deferred."""

    __slots__ = ("tensor", "_host")

    def __init__(self, tensor):
        self.tensor = tensor
        self._host = None

    def numpy(self):
        if self._host is None:
            self._host = self.tensor.cpu().numpy()
        return self._host

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, i):
        return self.numpy()[i]

    def __iter__(self):
        return iter(self.numpy())

    def __array__(self, dtype=None, copy=None):
        out = self.numpy()
        return out.astype(dtype) if dtype is not None else out

    def tolist(self):
        return self.numpy().tolist()


# ---------------------------------------------------------------------------
# phi recognition
# ---------------------------------------------------------------------------
class ScaleU8:
    """Explicit marker for the Atari feature extractor
    ``phi(x) = np.asarray(x, dtype=np.float32) / divisor``
    (examples/atari/train_dqn_batch_ale.py:229-231).  Callable on host data."""

    def __init__(self, divisor=255.0):
        self.divisor = float(divisor)

    def __call__(self, x):
        return np.asarray(x, dtype=np.float32) / np.float32(self.divisor)


def _scale_of(x, y):
    """d such that y == float32(x) / d elementwise (same shape, y float32), or None; an
    all-zero x is inconclusive and also gives None."""
    if y.shape != x.shape or y.dtype != np.float32:
        return None
    xf = x.astype(np.float32)
    nz = xf != 0
    if not np.any(nz):
        return None
    if np.array_equal(y, xf):
        return 1.0
    m = float(np.median(xf[nz] / y[nz]))
    for cand in (m, round(m)):
        if cand > 0 and np.array_equal(y, xf / np.float32(cand)):
            return float(cand)
    return None


def recognise_phi(phi, sample_obs):
    """Classify ``phi`` by evaluating it on a real observation and on a synthetic probe.

    Returns the divisor d such that phi(x) == float32(x) / d elementwise with
    the same shape (d == 1.0 covers cast-only and identity), or None if phi is
    something else (or cannot be decided yet).  Arbitrary Python callables cannot run
    on the device; the three forms used by the reference's example scripts can.

    One sample is not enough: an all-zero observation cannot tell x from x / 255, and a
    clipping or non-linear phi can act as a pure scale on it.  The probe covers the value
    range of the observation dtype; both have to agree."""
    if isinstance(phi, ScaleU8):
        return phi.divisor
    x = np.asarray(sample_obs)
    try:
        y = np.asarray(phi(sample_obs))
    except Exception:
        return None
    if y.shape != x.shape or y.dtype != np.float32:
        return None
    d = _scale_of(x, y)
    if d is None and np.any(x != 0):
        return None
    try:
        if np.issubdtype(x.dtype, np.integer):
            # the dtype's range, clamped to a 2^31 span (an int64 / uint64 span is 2^64 and
            # overflows the modulus; wide ranges say nothing more about a cast / scale phi)
            info = np.iinfo(x.dtype)
            lo = max(int(info.min), -(1 << 30))
            span = min(int(info.max), (1 << 30) - 1) - lo + 1
            probe = (np.arange(x.size, dtype=np.int64) * 7919 % span + lo).astype(
                x.dtype).reshape(x.shape)
        elif x.size == 1:
            probe = np.full(x.shape, 1000.0, dtype=x.dtype)
        else:
            probe = np.linspace(-1000.0, 1000.0, x.size).astype(x.dtype).reshape(x.shape)
        dp = _scale_of(probe, np.asarray(phi(probe)))
    except Exception:
        return d    # phi only accepts its own observation type: the sample decides
    if dp is None or (d is not None and d != dp):
        return None
    return dp
