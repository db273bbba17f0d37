"""Device-resident PrioritizedBuffer.

Mirrors ``pfrl.collections.prioritized.PrioritizedBuffer``
(/root/reference/pfrl/collections/prioritized.py:21-123): same constructor,
``append / popleft / sample / set_last_priority / __len__`` and the same
assertion behaviour, but the sum/min trees live in HBM as tagged nodes and are
driven by the kernels in pfrl_amd/csrc/sumtree.hip.  The host keeps only the
integer bookkeeping (TreeFrame) and the Python payload deque.
"""
import collections
import os

import numpy as np
import torch

from pfrl_amd import ops
from pfrl_amd._native import MAX_LEVELS, TreeDesc
from pfrl_amd.collections.tree_frame import TreeFrame, smax_log2_for_capacity
from pfrl_amd.staging import StagingRing, on_stream

TAG_ABSENT, TAG_PY, TAG_F32, TAG_F64 = 0, 1, 2, 3


def type_tag(x):
    """NEP-50 tag of a scalar priority."""
    if isinstance(x, np.float32):
        return TAG_F32
    if isinstance(x, np.float64):
        return TAG_F64
    if isinstance(x, (float, int)):
        return TAG_PY
    if isinstance(x, np.floating):
        return TAG_F64
    raise TypeError("unsupported priority type %r" % type(x))


class PrioritizedBuffer:
    def __init__(self, capacity=None, wait_priority_after_sampling=True,
                 initial_max_priority=1.0, device=None, max_size=None):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("pfrl_amd PrioritizedBuffer is device-resident (needs a GPU)")
        self.capacity = capacity
        self.wait_priority_after_sampling = wait_priority_after_sampling
        self.flag_wait_priority = False
        self.data = collections.deque()
        self.frame = TreeFrame()
        bound = capacity if capacity is not None else (max_size or (1 << 20))
        self._bound = bound
        self.log2_smax = smax_log2_for_capacity(bound)
        smax = 1 << self.log2_smax
        self._level_off = [0] * MAX_LEVELS
        off = 0
        for l in range(MAX_LEVELS):
            self._level_off[l] = off
            off += max(smax >> l, 1) if l <= self.log2_smax else 0
        n_nodes = off
        dev = self.device
        self.sum_val = torch.zeros(n_nodes, dtype=torch.float64, device=dev)
        self.sum_tag = torch.zeros(n_nodes, dtype=torch.uint8, device=dev)
        self.min_val = torch.zeros(n_nodes, dtype=torch.float64, device=dev)
        self.min_tag = torch.zeros(n_nodes, dtype=torch.uint8, device=dev)
        self._maxp_val = torch.full((1,), float(initial_max_priority), dtype=torch.float64,
                                    device=dev)
        self._maxp_tag = torch.full((1,), type_tag(initial_max_priority), dtype=torch.uint8,
                                    device=dev)
        self._stage = StagingRing(dev, slot_bytes=1 << 16, n_slots=64)
        self._pend_x, self._pend_v, self._pend_t, self._pend_m = [], [], [], []
        self._pend_at = {}      # leaf coordinate -> position in the pending lists
        self._sampled_x = None  # device int64 tensor of the last sample
        self._sample_out = {}
        # optional replay stream: all tree launches go there and the sample outputs
        # alternate between two buffer sets (see DeviceReplayStore.set_side_stream)
        self.side_stream = None
        self._sample_parity = 0
        self._n_sampled = 0
        # update_errors_device of the last minibatch, held back until the next leaf writes go
        # to the device (one launch and one path repair for both; see _launch_deferred)
        self._deferred = None
        self.defer_errors = os.environ.get("PFRL_TREE_FUSE_ERRORS", "1") != "0"
        self._desc = TreeDesc()
        d = self._desc
        d.sum_val, d.sum_tag = self.sum_val.data_ptr(), self.sum_tag.data_ptr()
        d.min_val, d.min_tag = self.min_val.data_ptr(), self.min_tag.data_ptr()
        d.maxp_val, d.maxp_tag = self._maxp_val.data_ptr(), self._maxp_tag.data_ptr()
        for l in range(MAX_LEVELS):
            d.level_off[l] = self._level_off[l]
        d.log2_smax = self.log2_smax

    # -- descriptor ---------------------------------------------------------
    def _sync_desc(self):
        d, f = self._desc, self.frame
        d.base, d.head, d.length, d.log2_size = f.base, f.head, f.length, f.log2_size
        for l in range(1, f.log2_size + 1):
            d.origin[l] = f.origin[l]
        return d

    def __len__(self):
        return len(self.data)

    # -- pending leaf writes ------------------------------------------------
    def _record(self, x, val, tag, use_maxp):
        # One launch applies all pending leaf writes concurrently: two writes to the same leaf
        # (an element appended and popped again before the next flush -- tiny capacities only)
        # would race, store by store.  Sequentially the later one wins: keep only that one.
        at = self._pend_at.get(x)
        if at is not None:
            self._pend_v[at], self._pend_t[at], self._pend_m[at] = val, tag, use_maxp
            return
        self._pend_at[x] = len(self._pend_x)
        self._pend_x.append(x)
        self._pend_v.append(val)
        self._pend_t.append(tag)
        self._pend_m.append(use_maxp)

    def _launch_deferred(self, desc, writes=None):
        """The held-back update_errors_device -- alone, or together with the leaf writes
        ``writes`` = (x, val, tag, use_maxp) when both fit one launch (returns True then).
        Both belong to the current frame: every frame change flushes first."""
        d, self._deferred = self._deferred, None
        if d is None:
            return False
        x, err, args, kw = d
        if writes is not None and x.numel() + writes[0].numel() <= 1024:
            ops.tree_update_errors_write_f32(desc, x, err, *args, writes=writes, **kw)
            return True
        ops.tree_update_errors_f32(desc, x, err, *args, **kw)
        return False

    def flush(self):
        """Launch the recorded leaf writes under the frame they belong to."""
        n = len(self._pend_x)
        if n == 0:
            if self._deferred is not None:
                with on_stream(self.side_stream):
                    self._launch_deferred(self._sync_desc())
            return
        desc = self._sync_desc()
        with on_stream(self.side_stream):
            for lo in range(0, n, 1024):
                hi = min(n, lo + 1024)
                x, v, t, m = self._stage.upload([
                    np.asarray(self._pend_x[lo:hi], dtype=np.int64),
                    np.asarray(self._pend_v[lo:hi], dtype=np.float64),
                    np.asarray(self._pend_t[lo:hi], dtype=np.uint8),
                    np.asarray(self._pend_m[lo:hi], dtype=np.uint8),
                ])
                if not self._launch_deferred(desc, (x, v, t, m)):
                    ops.tree_write(desc, x, v, t, m)
        self._pend_x, self._pend_v, self._pend_t, self._pend_m = [], [], [], []
        self._pend_at = {}

    def take_pending(self):
        """The recorded leaf writes as four arrays for a SHARED staging transfer (at most 1024
        of them; None otherwise) and the launch to make once they are on the device -- so that
        a sample's whole control traffic (these, the uniform draws, the store's new rows) crosses
        PCIe as ONE copy in front of the launches instead of one copy in front of each."""
        n = len(self._pend_x)
        if n == 0 or n > 1024:
            return None
        arrays = [np.asarray(self._pend_x, dtype=np.int64), np.asarray(self._pend_v, dtype=np.float64),
                  np.asarray(self._pend_t, dtype=np.uint8), np.asarray(self._pend_m, dtype=np.uint8)]
        desc = self._sync_desc()
        self._pend_x, self._pend_v, self._pend_t, self._pend_m = [], [], [], []
        self._pend_at = {}

        def launch(x, v, t, m):
            if not self._launch_deferred(desc, (x, v, t, m)):
                ops.tree_write(desc, x, v, t, m)

        return arrays, launch

    # -- reference API ------------------------------------------------------
    def append(self, value, priority=None):
        """prioritized.py:39-48"""
        if self.capacity is not None and len(self) == self.capacity:
            self.popleft()
        if self.frame.length >= self._bound and self.capacity is None:
            raise RuntimeError("unbounded PrioritizedBuffer exceeded max_size=%d" % self._bound)
        doubling = False
        if self.frame.will_change_on_append():
            self.flush()
            doubling = self.frame.length > 0
        x = self.frame.append()
        if doubling:
            self._clear_new_half()
        if priority is None:
            self._record(x, 0.0, TAG_PY, 1)
        else:
            self._record(x, float(priority), type_tag(priority), 0)
        self.data.append(value)

    def _clear_new_half(self):
        """The frame has just doubled (prioritized.py:214-220: the new root's right child is
        an EMPTY subtree).  On the device the nodes of that half are physical ring slots that
        an earlier incarnation of the frame may have used under other level origins and
        left behind with values -- ancestors above the then-current root are not repaired when
        their leaves are popped.  Mark every internal node of the new right half absent, so
        that path repairs of the leaves about to be appended see the empty siblings the
        reference has (found with an unbounded buffer and bursts of popleft, the prioritized
        episodic buffer's pattern: root sum off by a stale level-2 node)."""
        f = self.frame
        L = f.log2_size
        half = f.size // 2
        x0 = f.base + half
        smax = 1 << self.log2_smax
        with on_stream(self.side_stream):
            for l in range(1, L):
                M = max(smax >> l, 1)
                q0 = ((x0 - f.origin[l]) >> l) & (M - 1)
                cnt = min(half >> l, M)
                off = self._level_off[l]
                first = min(cnt, M - q0)
                for tags in (self.sum_tag, self.min_tag):
                    tags[off + q0:off + q0 + first].zero_()
                    if cnt > first:
                        tags[off:off + cnt - first].zero_()

    def popleft(self):
        """prioritized.py:50-54"""
        assert len(self) > 0
        x = self.frame.popleft_coord()
        self._record(x, 0.0, TAG_ABSENT, 0)
        if self.frame.will_change_on_popleft():
            self.flush()
        self.frame.popleft()
        return self.data.popleft()

    def next_appends_keep_frame(self, m=1):
        """True when the next ``m`` calls of ``append`` (each with the ``popleft`` a full buffer
        does first) leave the tree frame as it is, i.e. none of them flushes."""
        from pfrl_amd.collections.tree_frame import appends_keep_frame

        pops = self.capacity is not None and len(self) + m > self.capacity
        return appends_keep_frame(self.frame, m, pops)

    def sample_device(self, n, u01=None, normalize=1, beta=0.0, slot_mod=0, co_stage=None,
                      split=False):
        """Device-side ``sample``: returns a dict of device tensors (x, pri,
        pri_tag, prob, weight, total, total_tag, min_prob[, slot]).  ``u01``
        defaults to the draws np.random.uniform would consume (same stream).
        ``co_stage`` = (arrays, launch) of another object (the replay store's new rows) whose
        control traffic rides in the same host->device copy, or a callable returning that pair
        (or None), called after the preconditions hold.

        ``split=True``: only what does not depend on the PREVIOUS sample's priorities is done
        now -- the host side, the one staging transfer, the other object's launches -- and
        ``(out, finish)`` is returned; ``finish()`` launches the held-back priority update fused
        with the pending leaf writes and the draws.  A caller that knows the next sample point
        ahead of time (DQN._batch_observe_train_per) prepares it before it launches the update
        that produces those priorities, so that only two launches are left for afterwards."""
        assert split or not self.wait_priority_after_sampling or not self.flag_wait_priority
        assert len(self) >= n
        if callable(co_stage):
            # taken only now: take_pending() clears the other object's pending rows, which must
            # not be lost to a failed assert above
            co_stage = co_stage()
        pending = self.take_pending()
        if pending is None and not split:
            self.flush()            # (nothing pending, or more than one launch's worth)
        # (split with more than one launch's worth of writes: they stay recorded until finish(),
        # which launches them BEHIND the previous sample's priority update -- flushing here would
        # put the pops of the look-ahead appends in front of it, and the update would then write a
        # live priority onto a leaf that is already absent)
        if u01 is None:
            u01 = np.random.random_sample(n)
        key = n
        if self.side_stream is not None:
            self._sample_parity ^= 1
            key = (n, self._sample_parity)
        out = self._sample_out.get(key)
        if out is None:
            dev = self.device
            out = self._sample_out[key] = dict(
                x=torch.empty(n, dtype=torch.int64, device=dev),
                pri=torch.empty(n, dtype=torch.float64, device=dev),
                pri_tag=torch.empty(n, dtype=torch.uint8, device=dev),
                prob=torch.empty(n, dtype=torch.float64, device=dev),
                weight=torch.empty(n, dtype=torch.float32, device=dev),
                total=torch.empty(1, dtype=torch.float64, device=dev),
                total_tag=torch.empty(1, dtype=torch.uint8, device=dev),
                min_prob=torch.empty(1, dtype=torch.float64, device=dev),
                slot=torch.empty(n, dtype=torch.int32, device=dev),
            )
        out = dict(out)  # persistent buffers (stable addresses for graph replay)
        if not slot_mod:
            del out["slot"]
        with on_stream(self.side_stream):
            # ONE transfer for everything this sample's launches read from the host
            arrays = [np.asarray(u01, dtype=np.float64)]
            n_t = n_c = 0
            if pending is not None:
                arrays += pending[0]
                n_t = len(pending[0])
            if co_stage is not None:
                arrays += co_stage[0]
                n_c = len(co_stage[0])
            need = sum(((a.nbytes + 15) & ~15) for a in arrays)
            if need > self._stage.slot_bytes:
                # copies issued through the native h2d path (and side-stream kernels reading the
                # slots) may still be in flight: the old ring's pinned buffers stay alive
                self._retired_stages = getattr(self, "_retired_stages", []) + [self._stage]
                self._stage = StagingRing(self.device, slot_bytes=2 * need, n_slots=32)
            views = self._stage.upload(arrays)
            if co_stage is not None:
                co_stage[1](*views[1 + n_t:1 + n_t + n_c])

        def finish():
            assert not self.wait_priority_after_sampling or not self.flag_wait_priority
            with on_stream(self.side_stream):
                if self._fuse_into_sampler(n, n_t and views[1].numel(), bool(self._pend_x)):
                    # priorities of the last minibatch + the writes recorded since + these draws:
                    # ONE launch (nothing between the TD errors and the next minibatch but it)
                    d, self._deferred = self._deferred, None
                    dx, derr, dargs, dkw = d
                    ops.tree_update_errors_write_sample(
                        self._sync_desc(), dx, derr, *dargs, tuple(views[1:1 + n_t]) if n_t else None,
                        views[0], out, normalize, beta, slot_mod, **dkw)
                else:
                    if pending is not None:
                        pending[1](*views[1:1 + n_t])
                    elif self._pend_x or self._deferred is not None:
                        self.flush()        # (priority update first, fused with the first 1 024 writes)
                    ops.tree_sample(self._sync_desc(), views[0], out, normalize, beta, slot_mod)
            self._sampled_x = out["x"]
            self._n_sampled = n
            self.flag_wait_priority = True
            return out

        if split:
            return out, finish
        return finish()

    def _fuse_into_sampler(self, n_draws, n_writes, more_pending):
        """Can the held-back priority update (+ the staged writes) ride in the sampler's launch?
        (pfrl_tree_update_errors_write_sample: at most 64 leaves, the lean sampler's tree sizes;
        PFRL_TREE_FUSE_SAMPLE=0: separate launches)"""
        d = self._deferred
        if d is None or more_pending or os.environ.get("PFRL_TREE_FUSE_SAMPLE", "1") == "0":
            return False
        if os.environ.get("PFRL_TREE_SAMPLE") not in (None, "", "prefetch"):
            return False
        L = self.frame.log2_size
        return (n_draws >= 1 and d[0].numel() + int(n_writes or 0) <= 64 and 1 <= L <= 22
                and L - min(L, 9) + 1 <= 13)

    def sample(self, n, uniform_ratio=0):
        """prioritized.py:56-105 (host-visible results; one D2H sync).  ``uniform_ratio > 0``
        mixes in leaves picked uniformly (SumTreeQueue.uniform_sample, :278-292) and
        ``wait_priority_after_sampling=False`` puts every sampled priority back afterwards
        (remove=False, :289-291 / :308-310): both on the device trees, the probabilities of the
        mixed form evaluated on the host from the typed priorities (the reference's own
        expression on NumPy scalars: NEP-50 decides its precision)."""
        assert not self.wait_priority_after_sampling or not self.flag_wait_priority
        if uniform_ratio == 0 and self.wait_priority_after_sampling:
            out = self.sample_device(n)
            self._join()
            x = out["x"].cpu().numpy()
            idx = x - self.frame.head
            self.sampled_indices = [int(i) for i in idx]
            sampled = [self.data[i] for i in self.sampled_indices]
            probs = out["prob"].cpu().numpy().tolist()
            return sampled, probs, float(out["min_prob"].item())
        return self._sample_mixed(n, uniform_ratio)

    @staticmethod
    def _typed(v, t):
        """The NumPy / Python scalar a tagged tree value stands for."""
        return np.float32(v) if t == TAG_F32 else (np.float64(v) if t == TAG_F64 else float(v))

    def _sample_mixed(self, n, uniform_ratio):
        """_sample_indices_and_probabilities (prioritized.py:56-84) in the reference's order:
        total and minimum first, then the binomial split, sample_n_k and the uniform removals,
        then the prioritized draws -- all three consuming the global NumPy stream as there."""
        from pfrl_amd.utils.random import sample_n_k

        assert len(self) >= n
        stats = self.root_stats()                       # (flushes; the trees as sampling finds them)
        total = self._typed(*stats[0])
        min_prob = self._typed(*stats[1]) / total
        dev = self.device
        xs, pris = [], []
        n_pr = n
        if uniform_ratio > 0:
            n_uniform = np.random.binomial(n, uniform_ratio)
            un = np.asarray(list(sample_n_k(len(self), n_uniform)), dtype=np.int64)
            if n_uniform > 0:
                assert n_uniform <= 1024, "at most 1 024 uniform samples per call"
                with on_stream(self.side_stream):
                    (ux,) = self._stage.upload([un + self.frame.head])
                    ux = ux.clone()
                    ov = torch.empty(n_uniform, dtype=torch.float64, device=dev)
                    ot = torch.empty(n_uniform, dtype=torch.uint8, device=dev)
                    ops.tree_write_sum(self._sync_desc(), ux, None, None, ov, ot)
                    if not self.wait_priority_after_sampling:
                        # remove=False: uniform_sample itself puts them back (:289-291), i.e.
                        # BEFORE the prioritized draws, which may pick the same leaves again
                        ops.tree_write_sum(self._sync_desc(), ux, ov, ot)
                xs.append(ux)
                pris.append((ov, ot))
            n_pr = n - n_uniform
            min_prob = uniform_ratio / len(self) + (1 - uniform_ratio) * min_prob
        if n_pr > 0:
            self.flag_wait_priority = False             # (sample_device checks and sets it)
            out = self.sample_device(n_pr)
            xs.append(out["x"].clone())
            pris.append((out["pri"].clone(), out["pri_tag"].clone()))
        self._join()
        x_all = torch.cat(xs) if xs else torch.empty(0, dtype=torch.int64, device=dev)
        pv = torch.cat([p[0] for p in pris]).cpu().numpy() if pris else np.empty(0)
        pt = torch.cat([p[1] for p in pris]).cpu().numpy() if pris else np.empty(0, dtype=np.uint8)
        priorities = [self._typed(v, t) for v, t in zip(pv, pt)]
        probs = [uniform_ratio / len(self) + (1 - uniform_ratio) * pri / total for pri in priorities]
        if not self.wait_priority_after_sampling and n_pr > 0:
            # remove=False: prioritized_sample writes its leaves back after its draws (:308-310)
            with on_stream(self.side_stream):
                px, (ppv, ppt) = xs[-1], pris[-1]
                for lo in range(0, n_pr, 1024):
                    hi = min(n_pr, lo + 1024)
                    ops.tree_write_sum(self._sync_desc(), px[lo:hi].contiguous(),
                                       ppv[lo:hi].contiguous(), ppt[lo:hi].contiguous())
        self._sampled_x = x_all
        self._n_sampled = int(x_all.numel())
        self.sampled_indices = [int(i) for i in (x_all.cpu().numpy() - self.frame.head)]
        self.flag_wait_priority = True
        return [self.data[i] for i in self.sampled_indices], probs, min_prob

    def set_last_priority(self, priority):
        """prioritized.py:107-116 with host-side typed priorities."""
        assert not self.wait_priority_after_sampling or self.flag_wait_priority
        assert all([p > 0.0 for p in priority])
        assert self._n_sampled == len(priority)
        with on_stream(self.side_stream):
            v, t = self._stage.upload([
                np.asarray([float(p) for p in priority], dtype=np.float64),
                np.asarray([type_tag(p) for p in priority], dtype=np.uint8),
            ])
            ops.tree_set_priorities(self._sync_desc(), self._sampled_x, v, t, dedupe=True)
        self.flag_wait_priority = False
        self.sampled_indices = []
        self._n_sampled = 0

    def update_errors_device(self, err, error_min, pri_at_min, error_max, pri_at_max, eps, alpha,
                             pow_mode=0):
        """set_last_priority for f32 errors already on the device (DQN path)."""
        assert not self.wait_priority_after_sampling or self.flag_wait_priority
        assert self._n_sampled == err.numel()
        assert self._deferred is None
        args = (error_min, pri_at_min, error_max, pri_at_max, eps, alpha)
        kw = dict(dedupe=True, pow_mode=pow_mode)
        if self.defer_errors:
            # nothing reads the tree before the next flush / sample; the launch then carries
            # the appends recorded in between as well (err stays referenced until then)
            self._deferred = (self._sampled_x, err, args, kw)
        else:
            with on_stream(self.side_stream):
                ops.tree_update_errors_f32(self._sync_desc(), self._sampled_x, err, *args, **kw)
        self.flag_wait_priority = False
        self.sampled_indices = []
        self._n_sampled = 0

    # -- inspection (tests / statistics) ------------------------------------
    def _join(self):
        if self.side_stream is not None:
            self.side_stream.synchronize()

    @property
    def max_priority(self):
        if self._deferred is not None:
            self.flush()
        self._join()
        v = float(self._maxp_val.item())
        t = int(self._maxp_tag.item())
        return np.float32(v) if t == TAG_F32 else (np.float64(v) if t == TAG_F64 else v)

    def root_stats(self):
        """((sum, tag), (min, tag), (max_priority, tag)) read back from HBM."""
        self.flush()
        self._join()
        f = self.frame
        if f.length == 0:
            return None
        q = (f.base - f.origin[f.log2_size]) >> f.log2_size
        M = max((1 << self.log2_smax) >> f.log2_size, 1)
        i = self._level_off[f.log2_size] + (q & (M - 1))
        return ((float(self.sum_val[i].item()), int(self.sum_tag[i].item())),
                (float(self.min_val[i].item()), int(self.min_tag[i].item())),
                (float(self._maxp_val.item()), int(self._maxp_tag.item())))

    def dump_level(self, which, l):
        """Nodes of level ``l`` (0 = leaves) in frame order: (values, tags)."""
        self.flush()
        self._join()
        f = self.frame
        n = f.size >> l
        M = max((1 << self.log2_smax) >> l, 1)
        q0 = (f.base - f.origin[l]) >> l
        idx = (torch.arange(n, device=self.device) + q0) & (M - 1)
        idx = idx + self._level_off[l]
        val = (self.min_val if which else self.sum_val)[idx].cpu().numpy()
        tag = (self.min_tag if which else self.sum_tag)[idx].cpu().numpy()
        val = np.where(tag == 0, 0.0, val)
        return val, tag.astype(np.int32)


# The reference exposes its pointer-tree queues from this module; here the device trees are flat
# arrays in HBM (above), and these names are the host-side queues of the gpu=None path.
from pfrl_amd.collections.host_prioritized import (  # NOQA,E402
    _MinTreeQueue as MinTreeQueue, _SumTreeQueue as SumTreeQueue, _TreeQueue as TreeQueue)
