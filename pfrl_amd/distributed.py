"""Env-sharded data parallelism: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU path at all.  Here every
rank owns its slice of the environments, its own frame ring / replay store /
priority tree (per-GPU-local replay: no sample ever crosses GPUs) and the same
network replica; the only collective on the hot path is ONE all-reduce of the
flattened gradient per optimizer step (6.75 MB fp32 for the Nature DQN).  A
single flat bucket is used on purpose: xGMI is point-to-point, a ring over it
is per-link bound, and at this size the collective is latency dominated -- one
call is better than per-parameter buckets.
"""
import os
import weakref

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  Single process: (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PFRL_DIST_ALWAYS=1 initialises the group for a single rank too (a one-GPU box
    # can then exercise RCCL init, the per-update all-reduce and the split graph)
    always = os.environ.get("PFRL_DIST_ALWAYS") == "1"
    if (world > 1 or always) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PFRL_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if torch.cuda.is_available():
            idx = local % torch.cuda.device_count()
            torch.cuda.set_device(idx)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", idx)   # bind the communicator eagerly
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_envs(num_envs, rank=None, world=None):
    """Contiguous env slice [lo, hi) owned by ``rank``."""
    if world is None:
        world = world_size()
    if rank is None:
        rank = dist.get_rank() if world > 1 else 0
    assert num_envs % world == 0, "num_envs must be divisible by the number of GPUs"
    per = num_envs // world
    return rank * per, (rank + 1) * per


# reducers that take early per-tensor all-reduces; gradient producers call announce_grad()
_EARLY_REDUCERS = weakref.WeakSet()
# set by a graph capture that wants collectives captured with it (graphed_update.py)
_CAPTURE_COLLECTIVES = [False]


def announce_grad(param, grad):
    """Called by a gradient producer (pfrl_amd/nn/mfma_trunk.py backward) right after the
    launch that completes ``grad`` of ``param``: any data-parallel reducer that wants this
    tensor early starts its all-reduce now.  Free when no process group exists."""
    for r in list(_EARLY_REDUCERS):
        target = param
        if id(param) not in r._early:
            # the producer may hold another Python object for the same storage
            target = next((q for q in r._early.values() if q.data_ptr() == param.data_ptr()), None)
            if target is None:
                continue
        if r.grad_ready(target, grad):
            return True
    return False


class GradientAllReducer:
    """Average gradients across ranks: one flat all-reduce per step for the small tensors, and
    -- ``early_bytes`` -- separate, EARLY all-reduces for the few large ones.

    xGMI is point-to-point and a 6.75 MB ring all-reduce is latency / per-link bound, so the
    collective is kept off the critical path instead of being cut into many buckets: a
    parameter of at least ``early_bytes`` (the Nature network's 3136 x 512 layer is 6.4 of its
    6.75 MB, and its gradient is the first to come out of backward) is all-reduced on RCCL's
    own stream as soon as its gradient exists, while the rest of backward (the convolution
    gradients) still runs; everything else travels in one flat bucket at the end.  The
    gradient's producer announces it with :meth:`grad_ready` (the MFMA trunk's backward
    does, right after the launch that writes it); for ordinary autograd modules a
    post-accumulate hook does the same.  Results are identical to the single-bucket plan
    (same averaging, per tensor)."""

    def __init__(self, module, early_bytes=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._flat = None
        if early_bytes is None:
            early_bytes = int(os.environ.get("PFRL_EARLY_ALLREDUCE_BYTES", str(1 << 22)))
        self.early_bytes = early_bytes
        self._early = {id(p): p for p in self.params
                       if early_bytes > 0 and p.numel() * p.element_size() >= early_bytes}
        self._pending = {}      # id(param) -> (work handle or None, gradient tensor)
        self._hooks = []
        if self._early and self.active():
            # weak registration: a reducer lives as long as its agent does (evaluation copies,
            # re-created agents and tests must not leave reducers behind that announce_grad
            # would keep scanning), and the hooks hold the reducer weakly for the same reason
            _EARLY_REDUCERS.add(self)
            ref = weakref.ref(self)

            def on_accumulated(p, ref=ref):
                r = ref()
                if r is not None:
                    r._on_accumulated(p)

            for p in self._early.values():
                if hasattr(p, "register_post_accumulate_grad_hook"):
                    self._hooks.append(p.register_post_accumulate_grad_hook(on_accumulated))

    def close(self):
        """Detach from the early-announcement registry and remove the gradient hooks."""
        _EARLY_REDUCERS.discard(self)
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._pending = {}

    def __del__(self):
        try:
            for h in self._hooks:
                h.remove()
        except Exception:
            pass

    def active(self):
        return dist.is_available() and dist.is_initialized()

    # -- early (per-tensor) collectives ------------------------------------------------
    def _on_accumulated(self, p):
        if id(p) not in self._pending and p.grad is not None:
            self.grad_ready(p, p.grad)

    def grad_ready(self, param, grad):
        """``grad`` (the tensor that is, or is about to become, ``param.grad``) is complete:
        start its all-reduce now.  Ignored for parameters below ``early_bytes``, outside a
        process group, while a HIP graph is being captured without the collective in it, or
        when the gradient is going to be accumulated into an existing one."""
        if not self.active() or id(param) not in self._early or id(param) in self._pending:
            return False
        if param.grad is not None and param.grad is not grad:
            return False
        if grad.is_cuda and torch.cuda.is_current_stream_capturing() and not _CAPTURE_COLLECTIVES[0]:
            return False
        self._pending[id(param)] = (self._start(grad), grad)
        return True

    def _start(self, t):
        if dist.get_backend() == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def _finish_early(self):
        for work, t in self._pending.values():
            if work is not None:
                work.wait()
            if dist.get_backend() != "nccl":
                t.div_(world_size())
        self._pending = {}

    # -- the flat bucket ---------------------------------------------------------------
    def _views(self):
        grads = [p.grad for p in self.params
                 if p.grad is not None and id(p) not in self._pending]
        if not grads:
            return None, None
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        views = []
        off = 0
        for g in grads:
            views.append(self._flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        return grads, views

    # The three parts are separate so that a captured update can keep the two
    # multi-tensor copies inside its graphs and leave only the collective eager.
    def pack(self):
        """gradients -> flat bucket (one multi-tensor copy)."""
        if not self.active():
            return
        grads, views = self._views()
        if grads:
            torch._foreach_copy_(views, grads)

    def reduce_flat(self):
        """The collective itself: average the flat bucket over all ranks."""
        if not self.active() or self._flat is None:
            return
        if dist.get_backend() == "nccl":
            dist.all_reduce(self._flat, op=dist.ReduceOp.AVG)   # RCCL averages in the collective
        else:
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
            self._flat.div_(world_size())

    def unpack(self):
        """flat bucket -> gradients; joins the early all-reduces."""
        if not self.active():
            return
        grads, views = self._views()
        if grads:
            torch._foreach_copy_(grads, views)
        self._finish_early()

    def all_reduce(self):
        if not self.active():
            return
        self.pack()
        self.reduce_flat()
        self.unpack()

    def broadcast_parameters(self, module, src=0):
        if not (dist.is_available() and dist.is_initialized()):
            return
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def broadcast_agent(agent, src=0):
    """Make every replica start from rank ``src``'s weights: broadcasts the parameters and
    buffers of all ``nn.Module`` attributes an agent saves (online and target networks,
    normalisers, SAC's temperature).  Optimizer state is created lazily and therefore still
    empty at this point.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    seen = set()
    for name in getattr(agent, "saved_attributes", ()):
        module = getattr(agent, name, None)
        if isinstance(module, torch.nn.Module) and id(module) not in seen:
            seen.add(id(module))
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=src)


def global_mean_std(mean_std, n):
    """Combine per-rank (mean, biased std) over ``n`` local samples into the
    statistics of the union of all ranks' samples: one 3-scalar all-reduce
    (count, sum, sum of squares).  PPO standardises advantages over the whole
    dataset (reference ppo.py:476-478); with env sharding the dataset is the
    union of the shards.  No-op for a single process."""
    if world_size() == 1:
        return mean_std
    m, s = mean_std[0].double(), mean_std[1].double()
    acc = torch.stack([torch.as_tensor(float(n), dtype=torch.float64, device=mean_std.device),
                       m * n, (s * s + m * m) * n])
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    gm = acc[1] / acc[0]
    gv = torch.clamp(acc[2] / acc[0] - gm * gm, min=0.0)
    return torch.stack([gm, torch.sqrt(gv)]).to(mean_std.dtype)
