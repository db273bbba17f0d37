"""Env-sharded data parallelism: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU path at all.  Here every
rank owns its slice of the environments, its own frame ring / replay store /
priority tree (per-GPU-local replay: no sample ever crosses GPUs) and the same
network replica.  Per optimizer step the ranks exchange

* ONE all-reduce of a flat bucket holding every small gradient (the three convolutions and the
  head of the Nature DQN: 0.32 MB fp32) -- a single call on purpose: xGMI is point-to-point, a
  ring over it is per-link bound, and at this size the collective is latency dominated;
* for a large ``Linear`` layer at minibatch size (the 3136 x 512 layer: 6.4 of the network's
  6.75 MB) NOT its gradient but the two batch matrices the gradient is the product of: an
  all-gather of ``dy [M, F]`` and ``x [M, K]`` (0.46 MB per rank at M = 32) and a local
  ``dW = sum_g dy_g^T x_g / G`` -- 3.5 x fewer bytes on the links at G = 8 than all-reducing dW, and
  it can leave as soon as the layer's backward has run, under the convolution backward
  (``lowrank_pays``); a large gradient that is not such a product is all-reduced early instead.
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
            # the per-update collectives are driven directly on RCCL (pfrl_amd/rccl.py); the
            # process group is the control plane only, and on gloo there is no NCCL watchdog
            # thread in the process (its event polls abort HIP-graph captures on this stack)
            direct = os.environ.get("PFRL_RCCL_DIRECT", "1") != "0"
            backend = os.environ.get("PFRL_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() and not direct else "gloo")
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


def _is_dense(t):
    """Every element of the storage span exactly once (any permutation of a contiguous layout)."""
    if t.is_contiguous():
        return True
    order = sorted(range(t.dim()), key=lambda d: -t.stride(d))
    return t.permute(*order).is_contiguous()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def control_all_reduce_sum(t):
    """Control-plane SUM all-reduce of a small tensor, in place, on ANY backend: with gloo (the
    default control plane beside the direct RCCL data plane) a device tensor is staged through the
    host -- gloo's device support is a build option this code does not rely on."""
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def control_broadcast(t, src=0):
    """Control-plane broadcast of one tensor, in place, on any backend (see above)."""
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        if dist.get_rank() != src:
            t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


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


def lowrank_pays(M, F, K, world=None):
    """All-gathering the batch matrices of a Linear layer's weight gradient (every rank receives
    (G - 1) M (F + K) floats) instead of all-reducing the gradient itself (a ring moves
    2 (G - 1) / G F K floats per rank): only for minibatch-sized M."""
    world = world_size() if world is None else world
    if os.environ.get("PFRL_DP_LOWRANK") == "force":
        # (a one-GPU box exercising the exchange with a single-rank communicator)
        return M <= 256
    return world > 1 and M <= 256 and world * M * (F + K) <= F * K


def announce_lowrank(weight, bias, dy, x, mask=None, premasked=None):
    """Called by the producer of a large Linear layer's gradient INSTEAD of forming it: ``dy``
    [M, F] and ``x`` [M, K] are the layer's batch matrices (``mask``: the layer's output when its
    ReLU mask has not been applied to ``dy`` yet), dW = dy^T x, db = sum_m dy.  Returns True when a
    data-parallel reducer took them (it then owns ``weight.grad`` / ``bias.grad`` of this step);
    False = the caller forms the gradient."""
    for r in list(_EARLY_REDUCERS):
        if r.lowrank_ready(weight, bias, dy, x, mask, premasked=premasked):
            return True
    return False


def lowrank_wanted(weight, M):
    """Would :func:`announce_lowrank` be taken for this weight at batch ``M``?  (asked by the MFMA
    trunk's backward before it decides which launches to make)"""
    for r in list(_EARLY_REDUCERS):
        if r.wants_lowrank(weight, M):
            return True
    return False


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


def _lowrank_product(dy_all, x_all, weight, dw=None, db=None):
    """dW [F, K] = dy_all^T x_all in the layout of ``weight`` and db [F] = column sums of dy_all:
    the f32 MFMA weight-gradient program on the GPU (csrc/qnet.hip, the same kernel that forms the
    layer's local gradient; it writes the bias gradient beside it), a matmul elsewhere.  ``dw`` /
    ``db``: preallocated outputs (the caller's stream owns them)."""
    F, K = weight.shape
    M = dy_all.shape[0]
    if dy_all.is_cuda and K % 32 == 0 and F % 16 == 0 and weight.is_contiguous():
        from pfrl_amd import _native
        from pfrl_amd.nn import mfma_trunk as mt

        if _native.available():
            splits = mt._wgrad_splits(M, F, K)
            dev = dy_all.device
            dw = torch.empty((F, K), dtype=torch.float32, device=dev) if dw is None else dw
            db = torch.empty(F, dtype=torch.float32, device=dev) if db is None else db
            if splits == 1:
                _native.check(_native.lib().pfrl_conv2d_nhwc_bwd_weight(
                    mt._p(dy_all), None, mt._p(x_all), mt._p(dw), mt._p(db), 0, 0, M, 1, 1, K, F, 1, 1, 1,
                    1, mt._stream()), "lowrank_bwd_weight")
                return dw, db
            stride = F * K + F
            part = torch.empty(splits * stride, dtype=torch.float32, device=dev)
            _native.check(_native.lib().pfrl_conv2d_nhwc_bwd_weight(
                mt._p(dy_all), None, mt._p(x_all), mt._p(part), mt._p(part[F * K:]), stride, stride, M, 1,
                1, K, F, 1, 1, 1, splits, mt._stream()), "lowrank_bwd_weight")
            mt._reduce([(part, dw, None, stride, F * K, splits, 4, 0),
                        (part[F * K:], db, None, stride, F, splits, 4, 0)])
            return dw, db
    g = (dy_all.t() @ x_all).to(weight.dtype).reshape(weight.shape)
    s = dy_all.sum(dim=0)
    if dw is not None:
        dw.copy_(g)
        g = dw
    if db is not None:
        db.copy_(s)
        s = db
    return g, s


def _remember_input(module, args, output):
    if torch.is_grad_enabled() and args and args[0].dim() == 2:
        module._pfrl_dp_input = args[0].detach()


class _LowRankHook:
    """full-backward hook of an nn.Linear whose weight gradient is exchanged as (dy, x)."""

    def __init__(self, reducer_ref, module):
        self.ref, self.module = reducer_ref, module

    def __call__(self, module, grad_input, grad_output):
        r = self.ref()
        x = getattr(module, "_pfrl_dp_input", None)
        if r is None or x is None or grad_output[0] is None or grad_output[0].dim() != 2:
            return None
        module._pfrl_dp_input = None
        r.lowrank_ready(module.weight, module.bias, grad_output[0].detach(), x)
        return None


class _SideJoin:
    """wait() = the current stream waits for everything enqueued on the communicator's side
    stream so far (a stream wait: no host synchronisation, capturable)."""

    def __init__(self, side, device):
        self.side, self.device = side, device

    def wait(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.side:
            cur.wait_stream(self.side)


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
        self._comm = None           # pfrl_amd.rccl.Communicator for CUDA parameters (data plane)
        if self.params and self.params[0].is_cuda and dist.is_available() and dist.is_initialized():
            from pfrl_amd import rccl

            self._comm = rccl.default_comm(self.params[0].device)      # (collective on first use)
        self._flat_sources = None   # the bucket of pack_sources (padded segments, aliases .grad)
        self._bucket = None         # whichever bucket was packed last: what reduce_flat() reduces
        self._lowrank = []      # (weight, bias, dy_all, x_all, work handles, kept inputs)
        self._lowrank_modules = {}   # id(weight) -> nn.Linear whose hooks hand over (dy, x)
        # set by a caller whose optimizer can step single parameters (GraphedUpdate): called ON the
        # side stream with (weight, bias, dW, db) once the product exists; True = stepped there
        self.lowrank_step = None
        self._deferred = {}
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
            # plain nn.Linear layers with an early-sized weight hand over their batch matrices
            # when that is the cheaper exchange (autograd still forms dW; it is then not used)
            for m in module.modules():
                if type(m) is torch.nn.Linear and id(m.weight) in self._early:
                    self._lowrank_modules[id(m.weight)] = m
                    self._hooks.append(m.register_forward_hook(_remember_input))
                    self._hooks.append(m.register_full_backward_hook(_LowRankHook(ref, m)))

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
        if id(p) in self._pending or p.grad is None:
            return
        m = self._lowrank_modules.get(id(p))
        x = getattr(m, "_pfrl_dp_input", None) if m is not None else None
        if x is not None and self.wants_lowrank(p, x.shape[0]):
            # the layer's backward hook (which has dy) may fire after this one: the choice between
            # the two exchanges is made there, or by pack() if it never fires
            self._deferred[id(p)] = p
            return
        self.grad_ready(p, p.grad)

    def _start_deferred(self):
        for p in self._deferred.values():
            if id(p) not in self._pending and p.grad is not None:
                self.grad_ready(p, p.grad)
        self._deferred = {}

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

    # -- low-rank exchange of a large Linear layer's gradient ---------------------------------
    def _find_early(self, param):
        if id(param) in self._early:
            return self._early[id(param)]
        return next((q for q in self._early.values() if q.data_ptr() == param.data_ptr()), None)

    def wants_lowrank(self, weight, M):
        if not self.active() or os.environ.get("PFRL_DP_LOWRANK", "1") == "0" or weight.dim() != 2:
            return False
        target = self._find_early(weight)
        return (target is not None and id(target) not in self._pending
                and lowrank_pays(M, weight.shape[0], weight.shape[1]))

    def lowrank_ready(self, weight, bias, dy, x, mask=None, premasked=None):
        """Exchange this layer's gradient as its batch matrices, starting now: ``dy`` [M, F] (with
        ``mask``: the layer's output, its ReLU mask still to be applied), ``x`` [M, K].  On the GPU
        with the directly driven communicator EVERYTHING happens on its side stream, beside the
        backward launches that follow on the caller's: mask and 1/G scale, the two all-gathers,
        the product dW = sum_g dy_g^T x_g and the bias sums; :meth:`_finish_early` only joins the
        stream and hands the finished tensors over.  Same capture rule as :meth:`grad_ready`."""
        if not self.wants_lowrank(weight, dy.shape[0]):
            return False
        if dy.is_cuda and torch.cuda.is_current_stream_capturing() and not _CAPTURE_COLLECTIVES[0]:
            return False
        target = self._find_early(weight)
        if bias is not None:
            # (a producer may hold another Python object for the same storage)
            bias = next((q for q in self.params if q.data_ptr() == bias.data_ptr()), None)
        G = world_size()

        def prepared():
            if premasked is not None and premasked[1] == G:
                # (``premasked``: (dy masked and scaled by 1 / G already, G) -- the producer's own
                # launch wrote it, pfrl_dqn_head_td_loss)
                d = premasked[0]
            else:
                d = dy if mask is None else torch.ops.aten.threshold_backward(dy, mask, 0.0)
                if G > 1:
                    d = d * (1.0 / G)              # the average: dW = sum_g (dy_g / G)^T x_g
            d, xx = d.contiguous(), x.contiguous()
            return (d, xx, torch.empty((G * d.shape[0], d.shape[1]), dtype=d.dtype, device=d.device),
                    torch.empty((G * xx.shape[0], xx.shape[1]), dtype=xx.dtype, device=xx.device))

        self._pending[id(target)] = (None, None)     # (claims the parameter: no early all-reduce)
        self._deferred.pop(id(target), None)
        if self._comm is not None and dy.is_cuda:
            side = self._exchange_stream(dy.device)
            # outputs belong to the caller's stream (allocated before the fork), temporaries to the
            # side stream: no block changes streams while a kernel of the other may still use it
            dw = torch.empty(tuple(target.shape), dtype=torch.float32, device=dy.device)
            db = torch.empty(target.shape[0], dtype=torch.float32, device=dy.device)
            if side != torch.cuda.current_stream(dy.device):
                side.wait_stream(torch.cuda.current_stream(dy.device))
            with torch.cuda.stream(side):
                d, xx, dy_all, x_all = prepared()
                with self._comm.group():
                    self._comm.all_gather(dy_all, d, stream=side)
                    self._comm.all_gather(x_all, xx, stream=side)
                _lowrank_product(dy_all, x_all, target, dw, db)
                stepped = bool(self.lowrank_step is not None
                               and self.lowrank_step(target, bias, dw, db))
            self._lowrank.append((target, bias, (_SideJoin(side, dy.device),),
                                  (None, None) if stepped else (dw, db),
                                  (d, xx, dy_all, x_all, dy, x, mask, dw, db)))
            return True
        d, xx, dy_all, x_all = prepared()
        if dist.get_backend() == "nccl":
            works = (dist.all_gather_into_tensor(dy_all, d, async_op=True),
                     dist.all_gather_into_tensor(x_all, xx, async_op=True))
        elif d.is_cuda:
            # device batch matrices on gloo (direct data plane off or retired): via the host
            for full, part in ((dy_all, d), (x_all, xx)):
                h = torch.empty(full.shape, dtype=full.dtype)
                dist.all_gather(list(h.chunk(G)), part.detach().cpu())
                full.copy_(h)
            works = ()
        else:
            works = (dist.all_gather(list(dy_all.chunk(G)), d, async_op=True),
                     dist.all_gather(list(x_all.chunk(G)), xx, async_op=True))
        self._lowrank.append((target, bias, works, None, (d, xx, dy_all, x_all)))
        return True

    def _exchange_stream(self, device):
        """Where an early exchange is enqueued: the communicator's side stream (beside the
        backward launches that follow on the caller's) -- except inside a graph capture, where it
        stays ON the capturing stream.  Measured in round 5 with a live peer (two ranks, RCCL's
        socket transport; tools/rccl_multirank_check.py --graph-pattern A..F,
        profiles/r05_rccl_multirank.txt): collectives captured on the capture's origin stream
        replay correctly (one, several, grouped), while ANY collective on a stream that joined the
        capture through an event wait takes the process down with SIGSEGV inside
        hipStreamEndCapture -- grouped or not, with or without other collectives on the origin
        stream.  A captured update therefore gives up the overlap (PFRL_DP_FORK_IN_CAPTURE=1
        restores the fork for a stack where it works)."""
        cur = torch.cuda.current_stream(device)
        if (torch.cuda.is_current_stream_capturing()
                and os.environ.get("PFRL_DP_FORK_IN_CAPTURE", "0") != "1"):
            return cur
        return self._comm.side

    def _finish_lowrank(self):
        for weight, bias, works, done, keep in self._lowrank:
            for w in works:
                w.wait()
            dw, db = done if done is not None else _lowrank_product(keep[2], keep[3], weight)
            weight.grad = dw              # (None: already stepped on the side stream)
            if bias is not None:
                bias.grad = db
        self._lowrank = []

    def _averages(self, t):
        """Does the backend that will reduce ``t`` average by itself?"""
        return (self._comm is not None and t.is_cuda) or dist.get_backend() == "nccl"

    def _start(self, t):
        if self._comm is not None and t.is_cuda:
            # beside the compute stream: fork, enqueue, and let wait() join
            cur = torch.cuda.current_stream(t.device)
            side = self._exchange_stream(t.device)
            if side != cur:
                side.wait_stream(cur)
            self._comm.all_reduce(t, average=True, stream=side)
            return _SideJoin(side, t.device)
        if dist.get_backend() == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
        if t.is_cuda:
            # the process group carries a device gradient (direct data plane off or retired,
            # pfrl_amd/rccl.py): through the host, synchronously -- _finish_early divides
            control_all_reduce_sum(t)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def _finish_early(self):
        for work, t in self._pending.values():
            if work is not None:
                work.wait()
            if t is not None and not self._averages(t):
                t.div_(world_size())
        self._pending = {}
        self._finish_lowrank()

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
        self._start_deferred()
        grads, views = self._views()
        if grads:
            torch._foreach_copy_(views, grads)
        self._bucket = self._flat

    def current_bucket(self):
        return self._bucket

    def reduce_flat(self, bucket=None):
        """The collective itself: average the flat bucket over all ranks (``bucket``: the one a
        captured plan packed into; default = the one packed last)."""
        bucket = self._bucket if bucket is None else bucket
        if not self.active() or bucket is None:
            return
        if self._comm is not None and bucket.is_cuda:
            self._comm.all_reduce(bucket, average=True)     # stream-ordered on the current stream
        elif dist.get_backend() == "nccl":
            dist.all_reduce(bucket, op=dist.ReduceOp.AVG)   # RCCL averages in the collective
        else:
            control_all_reduce_sum(bucket)                  # (a device bucket on gloo: via the host)
            bucket.div_(world_size())

    # -- gradients that are still split-K slabs (the fused optimizer path) ----------------------
    def pack_sources(self, sources):
        """The data-parallel form of "the optimizer finishes the gradients" (GraphedUpdate): the
        backward pass left most small gradients as split-K slabs (``GradSource.slabs``) and never
        materialised them.  ONE multi-tensor fold launch sums every parameter's slabs straight
        into its segment of the flat bucket (the fold that would otherwise run inside the optimizer
        launch, writing where the collective reads: no pack copy), plain gradients are copied in
        beside them, and every such parameter's ``.grad`` becomes its (reduced-in-place) segment --
        so there is no unpack copy either.  Handled sources are removed from ``sources``."""
        if not self.active():
            return
        from pfrl_amd.nn import mfma_trunk as mt
        from pfrl_amd.optimizers import OPT_SLABS

        self._start_deferred()
        plan, n = [], 0
        for p in self.params:
            if id(p) in self._pending:
                continue
            src = sources.get(p)
            if src is not None and src.mode != OPT_SLABS:
                continue                    # (stepped already / not a gradient that crosses ranks)
            if src is None and p.grad is None:
                continue
            plan.append((p, src, n))
            n += (p.numel() + 3) & ~3       # 16-byte aligned segments (float4 stores of the fold)
        if not plan:
            self._bucket = None
            return
        dev = plan[0][0].device
        flat = self._flat_sources
        if flat is None or flat.numel() != n or flat.device != dev:
            flat = self._flat_sources = torch.zeros(n, dtype=torch.float32, device=dev)
        self._bucket = flat
        tasks, dst, srcs = [], [], []
        for p, src, off in plan:
            seg = flat[off:off + p.numel()]
            view = seg.as_strided(p.size(), p.stride())     # the parameter's own (dense) layout
            if src is not None:
                tasks.append((src.src, seg, None, src.stride, p.numel(), src.n_slabs, 4, 0))
                del sources[p]
            else:
                dst.append(view)
                srcs.append(p.grad)
            p.grad = view
        for i in range(0, len(tasks), 12):
            mt._reduce(tasks[i:i + 12])
        if dst:
            torch._foreach_copy_(dst, srcs)

    def finish_sources(self):
        """After :meth:`reduce_flat`: join the early all-reduces and low-rank exchanges (the flat
        segments ARE the gradients already)."""
        if self.active():
            self._finish_early()

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
        if self._all_reduce_in_bucket_views():
            return
        self.pack()
        self.reduce_flat()
        self.unpack()

    def _all_reduce_in_bucket_views(self):
        """pack -> all-reduce -> unpack for dense f32 device gradients WITHOUT the per-tensor
        copies: ONE multi-tensor launch moves every gradient into its 16-byte aligned segment of
        the flat bucket (raw memory order: the segment view takes the gradient's own strides), the
        collective reduces the bucket, and ``p.grad`` BECOMES the segment view -- no copy back.
        (``torch._foreach_copy_`` is a launch per tensor on this stack: 24 launches, 110 us of a
        1.7 ms PPO update at the 8-GPU rank's minibatch, profiles/r06_ppo_rank_shape.txt.)
        False = some gradient is outside what this covers; the caller takes the copy path."""
        params = [p for p in self.params if p.grad is not None and id(p) not in self._pending]
        if not params or not all(
                p.grad.is_cuda and p.grad.dtype == torch.float32 and p.grad.stride() == p.stride()
                and p.grad.data_ptr() % 16 == 0 and p.grad.numel() > 0 and _is_dense(p)
                for p in params):
            return False
        from pfrl_amd import _native

        if not _native.available() or os.environ.get("PFRL_DP_BUCKET_VIEWS", "1") == "0":
            return False
        from pfrl_amd.nn import mfma_trunk as mt

        self._start_deferred()
        params = [p for p in params if id(p) not in self._pending]
        n = sum((p.numel() + 3) & ~3 for p in params)
        dev = params[0].device
        flat = self._flat_sources
        if flat is None or flat.numel() != n or flat.device != dev:
            flat = self._flat_sources = torch.zeros(n, dtype=torch.float32, device=dev)
        tasks, off = [], 0
        for p in params:
            seg = flat[off:off + p.numel()]
            tasks.append((p.grad, seg, None, 0, p.numel(), 1, 4, 0))
            off += (p.numel() + 3) & ~3
        for i in range(0, len(tasks), 12):
            mt._reduce(tasks[i:i + 12])
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].as_strided(p.size(), p.stride())
            off += (p.numel() + 3) & ~3
        self._bucket = flat
        self.reduce_flat()
        self._finish_early()
        return True

    def broadcast_parameters(self, module, src=0):
        if not (dist.is_available() and dist.is_initialized()):
            return
        for t in list(module.parameters()) + list(module.buffers()):
            control_broadcast(t.data, src=src)


_CAPTURE_PROBE = {}


def captured_collectives_work(device, timeout_s=20.0):
    """Can an RCCL collective be captured in a HIP graph and replayed on THIS process group?
    Decided once per group by a probe instead of by hoping: a 1 024-element all-reduce is captured,
    replayed twice, its completion awaited by polling an event for at most ``timeout_s`` and its
    result checked; the verdict is then agreed between the ranks (MIN over an eager all-reduce), so
    that every rank builds the same plan.  A probe that hangs leaves that tiny graph behind and the
    caller on the split plan (graph -> eager collective -> graph), which needs no capture support."""
    import time

    from pfrl_amd import rccl

    if not (dist.is_available() and dist.is_initialized()):
        return False
    comm = rccl.default_comm(device)
    if comm is None and dist.get_backend() != "nccl":
        return False
    key = (dist.distributed_c10d._world.group_count, str(device), comm is not None)
    if key in _CAPTURE_PROBE:
        return _CAPTURE_PROBE[key]

    def all_reduce_sum(t):
        if comm is not None:
            comm.all_reduce(t, average=False)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    ok = 0.0
    try:
        world = dist.get_world_size()
        buf = torch.ones(1024, dtype=torch.float32, device=device)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            all_reduce_sum(buf)                              # communicator warm-up, eager
            buf.fill_(1.0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                all_reduce_sum(buf)
            done = torch.cuda.Event()
            g.replay()
            g.replay()
            done.record(side)
        t0 = time.time()
        while not done.query() and time.time() - t0 < timeout_s:
            time.sleep(0.002)
        if done.query():
            want = float(world) ** 2
            ok = 1.0 if bool((buf == want).all().item()) else 0.0
            torch.cuda.current_stream(device).wait_stream(side)
        # (timed out: the side stream is left alone -- waiting on it would hang this stream too)
    except Exception:            # capture refused, communicator error: the split plan it is
        ok = 0.0
    try:
        verdict = torch.tensor([ok], dtype=torch.float32)       # (control plane: a host tensor)
        if dist.get_backend() == "nccl":
            verdict = verdict.to(device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        ok = float(verdict.item())
    except Exception:
        ok = 0.0
    _CAPTURE_PROBE[key] = ok > 0.5
    return _CAPTURE_PROBE[key]


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
                control_broadcast(t.data, src=src)


def global_mean_std(mean_std, n):
    """Combine per-rank (mean, biased std) over ``n`` local samples into the
    statistics of the union of all ranks' samples: one 3-scalar all-reduce
    (count, sum, sum of squares).  PPO standardises advantages over the whole
    dataset (reference ppo.py:476-478); with env sharding the dataset is the
    union of the shards.  No-op for a single process."""
    if world_size() == 1 and not (os.environ.get("PFRL_DIST_ALWAYS") == "1" and dist.is_available()
                                  and dist.is_initialized()):
        return mean_std
    # (PFRL_DIST_ALWAYS=1: a one-GPU box pays for the exchange under its single-rank group, so
    # that a rank-shaped measurement -- bench.py also.*_rank_shape_g8 -- includes it)
    m, s = mean_std[0].double(), mean_std[1].double()
    acc = torch.stack([torch.as_tensor(float(n), dtype=torch.float64, device=mean_std.device),
                       m * n, (s * s + m * m) * n])
    control_all_reduce_sum(acc)
    gm = acc[1] / acc[0]
    gv = torch.clamp(acc[2] / acc[0] - gm * gm, min=0.0)
    return torch.stack([gm, torch.sqrt(gv)]).to(mean_std.dtype)
