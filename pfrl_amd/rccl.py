"""The data plane of env-sharded data parallelism: RCCL driven directly (ctypes on the ``librccl.so``
PyTorch ships), one communicator per process, collectives enqueued on HIP streams like any kernel.

Why not ``torch.distributed``'s nccl backend for the per-update exchange: every collective there is
a ``Work`` object that a watchdog thread polls with ``hipEventQuery``.  On this stack (ROCm 7.0.2,
PyTorch 2.10, RCCL 2.26) such a poll that lands while the main thread is capturing a HIP graph
raises ``hipErrorCapturedEvent`` inside the watchdog, which terminates the process -- measured in
round 4 on one GPU with a single-rank communicator: about every second ``bench.py`` run under a
process group aborted during its first captures, in the round-3 plan (eager all-reduce between two
graphs) as much as with the collective inside the graph (profiles/r04_dp_watchdog_crashes.txt).
Stream-ordered RCCL calls have no host-side completion object at all: nothing polls, a capture
records them like any other kernel launch, and a whole env range of updates -- collectives
included -- replays as ONE graph.

``torch.distributed`` stays the control plane (rendezvous, the unique id, parameter broadcast,
barriers, PPO's three-scalar statistics), on gloo by default so that no NCCL watchdog exists in
the process (``distributed.init_process_group_from_env``).

Nothing here is trusted before it has been seen to work on THIS set of ranks: the communicator is
created by a helper thread that the caller waits for with a deadline, then a self-check (an eager
all-reduce, an all-gather and a grouped pair of all-gathers on small buffers, completion polled
through an event, results compared) runs under a deadline too, and the ranks agree on the verdict
over the control plane.  Any failure -- an error code, a wrong result, a deadline -- retires the
direct data plane for the rest of the process: ``default_comm`` returns None, the reducers of
``pfrl_amd/distributed.py`` carry the gradients over the process group instead (staged through
the host on gloo: slow, but it has no way to fail that the rendezvous has not already survived),
the captured update falls back to graph -> eager collective -> graph, and :func:`status` says
``"fallback:<reason>"`` (``bench.py`` prints it as ``config.dp_plan``).

``PFRL_RCCL_SHARED_DEVICE=1`` (test boxes with ONE GPU): every rank poses as its own host
(``NCCL_HOSTID``), which gets several ranks on one device past RCCL's duplicate-GPU check; they
then talk over RCCL's socket transport on loopback.  Functionally a real multi-rank communicator
(unique-id exchange, ``ncclCommInitRank(nranks > 1)``, captured collectives, grouped all-gathers
with a live peer); says nothing about xGMI.
"""
import ctypes
import os
import threading
import time

import torch

NCCL_UNIQUE_ID_BYTES = 128
ncclSum, ncclAvg = 0, 4
ncclFloat32, ncclInt64 = 7, 4


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * NCCL_UNIQUE_ID_BYTES)]


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = ctypes.CDLL(path if os.path.exists(path) else "librccl.so")
        V, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(V), I, _UniqueId, I]
        L.ncclCommDestroy.argtypes = [V]
        L.ncclAllReduce.argtypes = [V, V, Z, I, I, V, V]
        L.ncclAllGather.argtypes = [V, V, Z, I, V, V]
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        L.ncclGetErrorString.argtypes = [I]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce",
                     "ncclAllGather", "ncclGroupStart", "ncclGroupEnd"):
            getattr(L, name).restype = I
        _LIB = L
    return _LIB


class DataPlaneError(RuntimeError):
    """An RCCL call returned an error code (or the communicator failed its start-up checks)."""


def _check(rc, what):
    if rc != 0:
        raise DataPlaneError("RCCL %s failed: %s" % (what, _lib().ncclGetErrorString(rc).decode()))


# why the direct data plane was retired in this process (None: it was not)
_RETIRED = [None]
_TIMINGS = {}


def retire(reason):
    """Give up the direct data plane for the rest of the process (every rank must call it, or
    none: the callers agree over the control plane first)."""
    if _RETIRED[0] is None:
        _RETIRED[0] = str(reason)
    for c in _COMM.values():
        c.abandon()
    _COMM.clear()


def status():
    """What carries the per-update exchange in this process, for bench lines and logs."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 and not _COMM:
        plan = "single-process"
    elif _RETIRED[0] is not None:
        plan = "fallback:%s" % _RETIRED[0]
    elif os.environ.get("PFRL_RCCL_DIRECT", "1") == "0":
        plan = "process-group:%s" % dist.get_backend()
    elif _COMM:
        plan = "rccl-direct"
    else:
        plan = "process-group:%s" % dist.get_backend()
    out = {"dp_plan": plan}
    out.update(_TIMINGS)
    return out


def _dtype(t):
    if t.dtype == torch.float32:
        return ncclFloat32
    if t.dtype == torch.int64:
        return ncclInt64
    raise TypeError("pfrl_amd.rccl: f32 / i64 tensors only, got %s" % t.dtype)


class Communicator:
    """One RCCL communicator over the ranks of the default process group."""

    def __init__(self, rank, world, device, unique_id, timeout_s=None):
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self._comm = ctypes.c_void_p()
        uid = _UniqueId()
        ctypes.memmove(ctypes.byref(uid), bytes(unique_id), NCCL_UNIQUE_ID_BYTES)
        if timeout_s is None:
            timeout_s = float(os.environ.get("PFRL_RCCL_INIT_TIMEOUT", "180"))
        box = {}

        def init():
            # (a blocking rendezvous between the ranks: on a helper thread, so that a peer that
            # never arrives costs a deadline, not the process)
            try:
                with torch.cuda.device(self.device):
                    box["rc"] = _lib().ncclCommInitRank(ctypes.byref(self._comm), world, uid, rank)
            except Exception as e:       # noqa: BLE001 -- reported below
                box["exc"] = e

        th = threading.Thread(target=init, name="pfrl-rccl-init", daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            self._comm = ctypes.c_void_p()      # (the thread may still write its own copy: abandoned)
            raise DataPlaneError("ncclCommInitRank did not return within %.0f s" % timeout_s)
        if "exc" in box:
            raise DataPlaneError("ncclCommInitRank raised %r" % (box["exc"],))
        _check(box["rc"], "CommInitRank")
        # exchanges that should run BESIDE compute (the low-rank all-gather under the convolution
        # backward) go to this stream, forked from / joined to the caller's with stream waits
        self.side = torch.cuda.Stream(self.device)

    def _stream(self, stream):
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        return ctypes.c_void_p(s.cuda_stream)

    def all_reduce(self, t, average=True, stream=None):
        """In place, on ``stream`` (default: the current one): stream-ordered, capturable."""
        assert t.is_cuda and t.is_contiguous()
        _check(_lib().ncclAllReduce(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()),
                                    t.numel(), _dtype(t), ncclAvg if average else ncclSum,
                                    self._comm, self._stream(stream)), "AllReduce")

    def all_gather(self, out, inp, stream=None):
        """out [world * n] <- every rank's inp [n], rank order."""
        assert out.is_cuda and inp.is_cuda and out.is_contiguous() and inp.is_contiguous()
        assert out.numel() == self.world * inp.numel() and out.dtype == inp.dtype
        _check(_lib().ncclAllGather(ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                    inp.numel(), _dtype(inp), self._comm, self._stream(stream)),
               "AllGather")

    def group(self):
        return _Group()

    def self_check(self, timeout_s=None):
        """Eager all-reduce (sum and average), all-gather and a grouped pair of all-gathers on
        small buffers, on the side stream; completion is polled through an event for at most
        ``timeout_s``.  Returns None when every result is right, else a short reason."""
        if timeout_s is None:
            timeout_s = float(os.environ.get("PFRL_RCCL_CHECK_TIMEOUT", "60"))
        dev, G, r = self.device, self.world, self.rank
        try:
            with torch.cuda.device(dev):
                self.side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(self.side):
                    a = torch.full((1024,), float(r + 1), device=dev)
                    b = torch.arange(256, dtype=torch.float32, device=dev) * (r + 1)
                    gi = torch.full((64,), float(r), device=dev)
                    go = torch.full((G * 64,), -1.0, device=dev)
                    g1, g2 = torch.full((G * 64,), -1.0, device=dev), torch.full((G * 1024,), -1.0, device=dev)
                    self.all_reduce(a, average=False, stream=self.side)
                    self.all_reduce(b, average=True, stream=self.side)
                    self.all_gather(go, gi, stream=self.side)
                    a2 = torch.full((1024,), float(r + 1), device=dev)
                    with self.group():
                        self.all_gather(g1, gi, stream=self.side)
                        self.all_gather(g2, a2, stream=self.side)
                    done = torch.cuda.Event()
                    done.record(self.side)
                t0 = time.time()
                while not done.query() and time.time() - t0 < timeout_s:
                    time.sleep(0.002)
                if not done.query():
                    return "self-check collectives did not complete within %.0f s" % timeout_s
                torch.cuda.current_stream(dev).wait_stream(self.side)
                ranks = torch.arange(G, dtype=torch.float32, device=dev)
                ok = (bool((a == G * (G + 1) / 2).all())
                      and bool(torch.allclose(b, torch.arange(256, dtype=torch.float32, device=dev) * ((G + 1) / 2),
                                              rtol=1e-6, atol=0))
                      and bool((go.view(G, 64) == ranks[:, None]).all())
                      and bool(torch.equal(g1, go))
                      and bool((g2.view(G, 1024) == (ranks[:, None] + 1)).all()))
                return None if ok else "self-check collectives returned wrong values"
        except Exception as e:       # noqa: BLE001
            return "self-check raised %s" % (str(e)[:200],)

    def destroy(self):
        if self._comm:
            _lib().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()

    def abandon(self):
        """Forget the communicator WITHOUT calling into RCCL (it failed a check or a deadline:
        ncclCommDestroy on it may block)."""
        self._comm = ctypes.c_void_p()


class _Group:
    def __enter__(self):
        _check(_lib().ncclGroupStart(), "GroupStart")

    def __exit__(self, *exc):
        _check(_lib().ncclGroupEnd(), "GroupEnd")


_COMM = {}


def default_comm(device):
    """The process's data-plane communicator (created on first use: a COLLECTIVE call -- every rank
    must reach it, which they do when they build their agent's GradientAllReducer).  None without a
    process group, on a CPU device, or with PFRL_RCCL_DIRECT=0."""
    import torch.distributed as dist

    device = torch.device(device)
    if (os.environ.get("PFRL_RCCL_DIRECT", "1") == "0" or device.type != "cuda"
            or not (dist.is_available() and dist.is_initialized()) or _RETIRED[0] is not None):
        return None
    # (one communicator per process-group generation: tests create and destroy groups in one process)
    key = (dist.distributed_c10d._world.group_count, dist.get_rank(), dist.get_world_size(),
           device.index)
    if key not in _COMM:
        destroy_all()
        rank, world = dist.get_rank(), dist.get_world_size()
        if os.environ.get("PFRL_RCCL_SHARED_DEVICE") == "1":
            # several ranks on ONE device (functional tests on a one-GPU box, see the module text)
            os.environ["NCCL_HOSTID"] = "pfrl-shared-device-rank-%d" % rank
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        # the HIP context of this process exists before RCCL is entered (on a helper thread, below):
        # a rank whose first GPU call was ncclCommInitRank died with SIGSEGV inside RCCL's
        # topology discovery (round 5, tools/rccl_multirank_check.py on a fresh process)
        with torch.cuda.device(device):
            torch.zeros(1, device=device)
            torch.cuda.synchronize(device)
        t0 = time.time()
        box, comm, why = [None], None, None
        if rank == 0:
            try:
                uid = _UniqueId()
                _check(_lib().ncclGetUniqueId(ctypes.byref(uid)), "GetUniqueId")
                box[0] = ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)
            except Exception as e:       # noqa: BLE001 -- every rank learns of it below
                box[0] = "error: %s" % (str(e)[:200],)
        dist.broadcast_object_list(box, src=0)          # control plane: any backend
        if isinstance(box[0], str):
            why = box[0]
        else:
            try:
                comm = Communicator(rank, world, device, box[0])
            except DataPlaneError as e:
                why = str(e)[:200]
        _TIMINGS["rccl_init_s"] = round(time.time() - t0, 3)
        if comm is not None:
            t0 = time.time()
            why = comm.self_check()
            _TIMINGS["rccl_self_check_s"] = round(time.time() - t0, 3)
        # one verdict for all ranks (a rank that is fine must not keep a communicator whose peers
        # have walked away from it)
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        bad = [(r, w) for r, w in enumerate(reasons) if w is not None]
        if bad:
            if comm is not None:
                comm.abandon()
            retire("rank %d: %s" % bad[0])
            return None
        _COMM[key] = comm
    return _COMM[key]


def destroy_all():
    for c in _COMM.values():
        c.destroy()
    _COMM.clear()
