"""Pinned host staging for small host->device control traffic.

The replay path keeps all payload in HBM; what crosses PCIe per call is a few
hundred bytes of indices / random draws / metadata.  A ring of pinned buffers
with one event per slot lets the host run ahead of the device without ever
overwriting a buffer whose async copy has not executed yet.
"""
import numpy as np
import torch


class StagingRing:
    # seconds the host spent waiting for a slot's previous transfer (all rings): how far ahead of
    # the device the host runs shows up here, not in its own work
    wait_s = 0.0

    def __init__(self, device, slot_bytes=1 << 16, n_slots=64):
        self.device = torch.device(device)
        self.slot_bytes = slot_bytes
        self.n_slots = n_slots
        self._host = [torch.empty(slot_bytes, dtype=torch.uint8).pin_memory()
                      for _ in range(n_slots)]
        self._host_np = [h.numpy() for h in self._host]
        self._host_ptr = [h.data_ptr() for h in self._host]
        self._dev = [torch.empty(slot_bytes, dtype=torch.uint8, device=self.device)
                     for _ in range(n_slots)]
        self._dev_ptr = [d.data_ptr() for d in self._dev]
        # one event object per slot, re-recorded on every use; typed views of the device slot
        # cached by layout (the same few layouts come back on every call)
        self._events = [torch.cuda.Event() for _ in range(n_slots)]
        self._used = [False] * n_slots
        self._views = [dict() for _ in range(n_slots)]
        self._i = 0
        self._copy = None

    def _wait(self, i):
        if self._used[i]:
            ev = self._events[i]
            if not ev.query():
                import time

                t0 = time.perf_counter()
                ev.synchronize()
                StagingRing.wait_s += time.perf_counter() - t0

    def _ship(self, i, total):
        """The first ``total`` bytes of pinned slot i -> device slot i on the current stream."""
        if self._copy is None:
            import ctypes

            from pfrl_amd import _native

            self._copy = _native.lib().pfrl_h2d_async
            self._vp = ctypes.c_void_p
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        st = raw(self.device.index if self.device.index is not None else torch.cuda.current_device()) \
            if raw is not None else torch.cuda.current_stream(self.device).cuda_stream
        rc = self._copy(self._dev_ptr[i], self._host_ptr[i], total, self._vp(st))
        if rc != 0:
            raise RuntimeError("pfrl_h2d_async failed (%d)" % rc)
        self._events[i].record()
        self._used[i] = True

    def reserve(self):
        """Next pinned slot for the caller to fill in place: (numpy uint8 view, token).  Waits
        for the slot's previous transfer.  Follow with :meth:`commit`."""
        i = self._i
        self._i = (i + 1) % self.n_slots
        self._wait(i)
        return self._host_np[i], i

    def commit(self, token, nbytes):
        """Ship the first ``nbytes`` of a reserved slot in ONE async transfer; returns the
        device uint8 buffer of that slot (valid until the ring wraps)."""
        self._ship(token, (int(nbytes) + 15) & ~15)
        return self._dev[token]

    def commit_to(self, token, nbytes, dst):
        """:meth:`commit` with the bytes shipped into ``dst`` (a device uint8 tensor that keeps its
        address, e.g. the input block of a captured graph) instead of the ring's device slot."""
        dev_ptr = self._dev_ptr[token]
        self._dev_ptr[token] = dst.data_ptr()
        try:
            self._ship(token, (int(nbytes) + 15) & ~15)
        finally:
            self._dev_ptr[token] = dev_ptr
        return dst

    def upload_to(self, dst, arrays):
        """:meth:`upload` with the bytes shipped into ``dst`` (a device uint8 tensor that keeps its
        address: the input block of a captured graph); arrays are laid out back to back at
        16-byte aligned offsets, as :meth:`upload` lays them out."""
        i = self._i
        self._i = (i + 1) % self.n_slots
        self._wait(i)
        hb = self._host_np[i]
        off = 0
        for a in arrays:
            a = np.ascontiguousarray(a)
            off = (off + 15) & ~15
            nb = a.nbytes
            if off + nb > self.slot_bytes or off + nb > dst.numel():
                raise ValueError("staging slot / destination too small: need %d bytes" % (off + nb))
            hb[off:off + nb] = a.view(np.uint8).reshape(-1)
            off += nb
        return self.commit_to(i, off, dst)

    @staticmethod
    def view(dev, offset, count, dtype, shape=None):
        """Typed view of ``count`` elements at byte ``offset`` of a staged device buffer."""
        t = dev[offset:offset + count * dtype.itemsize].view(dtype)
        return t if shape is None else t.view(shape)

    def upload(self, arrays):
        """Copy a list of numpy arrays to the device in ONE async transfer.
        Returns device tensors viewing the staged bytes (valid until the ring
        wraps, i.e. for the next ``n_slots - 1`` uploads)."""
        i = self._i
        self._i = (i + 1) % self.n_slots
        self._wait(i)
        hb = self._host_np[i]
        off = 0
        key = []
        for a in arrays:
            a = np.ascontiguousarray(a)
            nb = a.nbytes
            off = (off + 15) & ~15
            if off + nb > self.slot_bytes:
                raise ValueError("staging slot too small: need %d bytes" % (off + nb))
            hb[off:off + nb] = a.view(np.uint8).reshape(-1)
            key.append((off, a.dtype.num, a.shape))
            off += nb
        self._ship(i, (off + 15) & ~15)
        key = tuple(key)
        outs = self._views[i].get(key)
        if outs is None:
            dev = self._dev[i]
            outs = []
            for (o, _, shape), a in zip(key, arrays):
                a = np.asarray(a)
                outs.append(dev[o:o + a.nbytes].view(_TORCH_DTYPES[a.dtype]).view(shape))
            if len(self._views[i]) > 32:
                self._views[i].clear()
            self._views[i][key] = outs
        return list(outs)


_TORCH_DTYPES = {
    np.dtype(np.uint8): torch.uint8,
    np.dtype(np.int32): torch.int32,
    np.dtype(np.int64): torch.int64,
    np.dtype(np.float32): torch.float32,
    np.dtype(np.float64): torch.float64,
    np.dtype(np.bool_): torch.bool,
}


class _NullScope:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_SCOPE = _NullScope()


def on_stream(stream):
    """``with on_stream(s):`` -- launches inside go to HIP stream ``s``; a no-op
    scope when ``s`` is None (everything stays on the caller's stream)."""
    return _NULL_SCOPE if stream is None else torch.cuda.stream(stream)
