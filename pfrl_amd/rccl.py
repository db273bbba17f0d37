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
"""
import ctypes
import os

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


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("RCCL %s failed: %s" % (what, _lib().ncclGetErrorString(rc).decode()))


def _dtype(t):
    if t.dtype == torch.float32:
        return ncclFloat32
    if t.dtype == torch.int64:
        return ncclInt64
    raise TypeError("pfrl_amd.rccl: f32 / i64 tensors only, got %s" % t.dtype)


class Communicator:
    """One RCCL communicator over the ranks of the default process group."""

    def __init__(self, rank, world, device, unique_id):
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self._comm = ctypes.c_void_p()
        uid = _UniqueId()
        ctypes.memmove(ctypes.byref(uid), bytes(unique_id), NCCL_UNIQUE_ID_BYTES)
        with torch.cuda.device(self.device):
            _check(_lib().ncclCommInitRank(ctypes.byref(self._comm), world, uid, rank), "CommInitRank")
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

    def destroy(self):
        if self._comm:
            _lib().ncclCommDestroy(self._comm)
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
            or not (dist.is_available() and dist.is_initialized())):
        return None
    # (one communicator per process-group generation: tests create and destroy groups in one process)
    key = (dist.distributed_c10d._world.group_count, dist.get_rank(), dist.get_world_size(),
           device.index)
    if key not in _COMM:
        destroy_all()
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:
            uid = _UniqueId()
            _check(_lib().ncclGetUniqueId(ctypes.byref(uid)), "GetUniqueId")
            box[0] = ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)
        dist.broadcast_object_list(box, src=0)          # control plane: any backend
        _COMM[key] = Communicator(rank, world, device, box[0])
    return _COMM[key]


def destroy_all():
    for c in _COMM.values():
        c.destroy()
    _COMM.clear()
