"""Pinned host staging for small host->device control traffic.

The replay path keeps all payload in HBM; what crosses PCIe per call is a few
hundred bytes of indices / random draws / metadata.  A ring of pinned buffers
with one event per slot lets the host run ahead of the device without ever
overwriting a buffer whose async copy has not executed yet.
"""
import numpy as np
import torch


class StagingRing:
    def __init__(self, device, slot_bytes=1 << 16, n_slots=64):
        self.device = torch.device(device)
        self.slot_bytes = slot_bytes
        self.n_slots = n_slots
        self._host = [torch.empty(slot_bytes, dtype=torch.uint8).pin_memory()
                      for _ in range(n_slots)]
        self._host_np = [h.numpy() for h in self._host]
        self._dev = [torch.empty(slot_bytes, dtype=torch.uint8, device=self.device)
                     for _ in range(n_slots)]
        self._events = [None] * n_slots
        self._i = 0

    def reserve(self):
        """Next pinned slot for the caller to fill in place: (numpy uint8 view, token).  Waits
        for the slot's previous transfer.  Follow with :meth:`commit`."""
        i = self._i
        self._i = (i + 1) % self.n_slots
        ev = self._events[i]
        if ev is not None:
            ev.synchronize()
            self._events[i] = None
        return self._host_np[i], i

    def commit(self, token, nbytes):
        """Ship the first ``nbytes`` of a reserved slot in ONE async transfer; returns the
        device uint8 buffer of that slot (valid until the ring wraps)."""
        i = token
        total = (int(nbytes) + 15) & ~15
        dev = self._dev[i]
        dev[:total].copy_(self._host[i][:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[i] = ev
        return dev

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
        ev = self._events[i]
        if ev is not None:
            ev.synchronize()
        hb = self._host_np[i]
        off = 0
        spans = []
        for a in arrays:
            a = np.ascontiguousarray(a)
            nb = a.nbytes
            off = (off + 15) & ~15
            if off + nb > self.slot_bytes:
                raise ValueError("staging slot too small: need %d bytes" % (off + nb))
            hb[off:off + nb] = a.view(np.uint8).reshape(-1)
            spans.append((off, nb, a.dtype, a.shape))
            off += nb
        total = (off + 15) & ~15
        dev = self._dev[i]
        dev[:total].copy_(self._host[i][:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[i] = ev
        outs = []
        for (o, nb, dt, shape) in spans:
            t = dev[o:o + nb].view(_TORCH_DTYPES[np.dtype(dt)]).view(shape)
            outs.append(t)
        return outs


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
