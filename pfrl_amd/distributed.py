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


class GradientAllReducer:
    """Average gradients across ranks with one flat all-reduce per step."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._flat = None

    def active(self):
        return dist.is_available() and dist.is_initialized()

    def _views(self):
        grads = [p.grad for p in self.params if p.grad is not None]
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
        """flat bucket -> gradients."""
        if not self.active():
            return
        grads, views = self._views()
        if grads:
            torch._foreach_copy_(grads, views)

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
