"""Optimizers whose step is one fused HIP launch.

``FusedRMSprop`` is ``torch.optim.RMSprop`` (same constructor, same state dict
layout, same arithmetic as its foreach implementation) with ``step()`` replaced
by pfrl_rmsprop_step: one multi-tensor kernel for all parameters.  It is what
examples/atari/train_dqn_batch_ale.py:199-206 constructs, minus ~70 us of tiny
foreach kernels per update.  Falls back to torch's own step for configurations
the kernel does not cover (momentum > 0, maximize, non-f32, CPU tensors).
"""
import ctypes

import torch

from pfrl_amd import _native


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


OPT_PLAIN, OPT_SLABS, OPT_LOWRANK, OPT_LOWRANK_BIAS, OPT_FOLD = 0, 1, 2, 3, 4   # PFRL_OPT_*
OPT_DONE = -1     # (host side only: GradSource.done())


class GradSource:
    """A gradient in the form the backward pass left it (see ``pfrl_rmsprop_fused_step``):
    ``slabs(part, stride, n)`` = split-K partial slabs still to be summed; ``lowrank(dy, mask, x)``
    = the weight of a Linear layer as the product dy^T x of its batch matrices; ``lowrank_bias(dy,
    mask)`` = that layer's bias.  Tensors are kept alive by the source."""

    __slots__ = ("mode", "src", "mask", "x", "stride", "n_slabs", "M", "F", "K")

    def __init__(self, mode, src, mask=None, x=None, stride=0, n_slabs=0, M=0, F=0, K=0):
        self.mode, self.src, self.mask, self.x = mode, src, mask, x
        self.stride, self.n_slabs, self.M, self.F, self.K = stride, n_slabs, M, F, K

    @classmethod
    def slabs(cls, part, stride, n_slabs):
        return cls(OPT_SLABS, part, stride=int(stride), n_slabs=int(n_slabs))

    @classmethod
    def lowrank(cls, dy, mask, x):
        M, F = dy.shape
        return cls(OPT_LOWRANK, dy, mask=mask, x=x, M=int(M), F=int(F), K=int(x.shape[1]))

    @classmethod
    def lowrank_bias(cls, dy, mask):
        M, F = dy.shape
        return cls(OPT_LOWRANK_BIAS, dy, mask=mask, M=int(M), F=int(F))

    @classmethod
    def done(cls):
        """The parameter was already stepped in this update (its RMSprop step rode in a backward
        launch, ``FusedRMSprop.ride_arrays``): the optimizer launch skips it."""
        return cls(OPT_DONE, None)

    @staticmethod
    def lowrank_supported(M, F, K):
        return K % 64 == 0 and F % 16 == 0 and M % 4 == 0 and 4 <= M <= 32


class FusedRMSprop(torch.optim.RMSprop):
    def _fusable(self, group):
        return (group["momentum"] == 0 and not group.get("maximize", False)
                and not group.get("differentiable", False))

    # -- the step that finishes the gradients (csrc/optim.hip k_rmsprop_fused) ---------------
    def accepts_sources(self):
        """One parameter group the kernel covers: the backward pass may then hand over
        gradients as :class:`GradSource` s instead of materialising them."""
        return len(self.param_groups) == 1 and self._fusable(self.param_groups[0])

    def ride_arrays(self, pairs):
        """Arguments of ``pfrl_conv2d_nhwc_bwd_weight_ride`` for ``pairs`` = [(parameter, finished
        gradient tensor)]: the steps of these parameters run inside a backward launch instead of
        the optimizer's own (``GradSource.done()`` then tells ``step_from_sources`` to skip them).
        None when a tensor does not meet the kernel's layout requirements."""
        if not self.accepts_sources() or not 1 <= len(pairs) <= 4:
            return None
        group = self.param_groups[0]
        centered = bool(group["centered"])
        ptrs = []
        for p, g in pairs:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device)
                              if group.get("capturable", False) else torch.tensor(0.0))
                st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if centered:
                    st["grad_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            sq = st["square_avg"]
            ga = st["grad_avg"] if centered else None
            ts = [p, g, sq] + ([ga] if centered else [])
            if not all(t.is_cuda and t.dtype == torch.float32 and _dense(t)
                       and t.stride() == p.stride() and t.data_ptr() % 16 == 0 for t in ts):
                return None
            if p.numel() % 4 != 0 or p.numel() == 0:
                return None
            ptrs.append((p.data_ptr(), g.data_ptr(), sq.data_ptr(), ga.data_ptr() if centered else 0,
                         p.numel()))
        n = len(ptrs)
        V = ctypes.c_void_p * n
        return (n, V(*[t[0] for t in ptrs]), V(*[t[1] for t in ptrs]), V(*[t[2] for t in ptrs]),
                V(*[t[3] for t in ptrs]), (ctypes.c_int64 * n)(*[t[4] for t in ptrs]),
                float(group["lr"]), float(group["alpha"]), float(group["eps"]),
                float(group["weight_decay"]), int(centered))

    def ride_set(self, entries):
        """Hand the steps of ``entries`` = [(parameter, gradient)] -- gradient: a finished dense
        tensor or a ``GradSource.slabs`` -- to the NEXT backward launch of this thread
        (``pfrl_ride_set``: consumed by ``pfrl_conv2d_nhwc_bwd`` / ``..._bwd_weight_ride``, which run
        them as extra workgroups).  True = set; the caller then marks the parameters
        ``GradSource.done()``.  False = a tensor is outside what the kernel covers (nothing set)."""
        if not self.accepts_sources() or not 1 <= len(entries) <= 8:
            return False
        group = self.param_groups[0]
        centered = bool(group["centered"])
        rows = []
        for p, g in entries:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device)
                              if group.get("capturable", False) else torch.tensor(0.0))
                st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if centered:
                    st["grad_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            sq = st["square_avg"]
            ga = st["grad_avg"] if centered else None
            if isinstance(g, GradSource):
                if g.mode != OPT_SLABS or g.stride % 4 != 0:
                    return False
                src, n_slabs, stride = g.src, g.n_slabs, g.stride
            else:
                if g.stride() != p.stride() or g.dtype != torch.float32:
                    return False
                src, n_slabs, stride = g, 0, 0
            ts = [p, sq] + ([ga] if centered else [])
            if not all(t.is_cuda and t.dtype == torch.float32 and _dense(t) and t.stride() == p.stride()
                       and t.data_ptr() % 16 == 0 for t in ts) or src.data_ptr() % 16 != 0:
                return False
            if p.numel() == 0:
                return False
            rows.append((p.data_ptr(), src.data_ptr(), sq.data_ptr(), ga.data_ptr() if centered else 0,
                         p.numel(), n_slabs, stride))
        n = len(rows)
        V = ctypes.c_void_p * n
        rc = _native.lib().pfrl_ride_set(
            n, V(*[r[0] for r in rows]), V(*[r[1] for r in rows]), V(*[r[2] for r in rows]),
            V(*[r[3] for r in rows]), (ctypes.c_int64 * n)(*[r[4] for r in rows]),
            (ctypes.c_int32 * n)(*[r[5] for r in rows]), (ctypes.c_int64 * n)(*[r[6] for r in rows]),
            float(group["lr"]), float(group["alpha"]), float(group["eps"]),
            float(group["weight_decay"]), int(centered))
        return rc == 0

    def ride_clear(self):
        _native.lib().pfrl_ride_set(0, None, None, None, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0)

    def step_from_sources(self, sources, folds=()):
        """``step()`` where the gradient of parameter ``p`` is ``sources[p]`` (a GradSource) if
        present and ``p.grad`` otherwise; ``folds`` = (part, out, stride, n_slabs) slab sums with
        no parameter behind them (loss terms), finished by the same launch."""
        group = self.param_groups[0]
        centered = bool(group["centered"])
        tasks = []
        keep = []
        for p in group["params"]:
            src = sources.get(p)
            if (src is None and p.grad is None) or (src is not None and src.mode == OPT_DONE):
                continue
            st = self.state[p]
            if len(st) == 0:
                st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device)
                              if group.get("capturable", False) else torch.tensor(0.0))
                st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if centered:
                    st["grad_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            assert p.is_cuda and p.dtype == torch.float32 and _dense(p) \
                and st["square_avg"].stride() == p.stride()
            t = _native.OptTask()
            t.p, t.sq = p.data_ptr(), st["square_avg"].data_ptr()
            t.ga = st["grad_avg"].data_ptr() if centered else None
            t.numel = p.numel()
            if src is None:
                g = p.grad
                assert g.dtype == torch.float32 and g.stride() == p.stride()
                t.mode, t.src = OPT_PLAIN, g.data_ptr()
            else:
                t.mode, t.src = src.mode, src.src.data_ptr()
                t.mask = src.mask.data_ptr() if src.mask is not None else None
                t.x = src.x.data_ptr() if src.x is not None else None
                t.slab_stride, t.n_slabs = src.stride, src.n_slabs
                t.M, t.F, t.K = src.M, src.F, src.K
                if src.mode in (OPT_LOWRANK, OPT_LOWRANK_BIAS):
                    assert p.is_contiguous()
                keep.append(src)
            tasks.append(t)
        for part, out, stride, n_slabs in folds:
            t = _native.OptTask()
            t.mode, t.src, t.out = OPT_FOLD, part.data_ptr(), out.data_ptr()
            t.numel, t.slab_stride, t.n_slabs = out.numel(), int(stride), int(n_slabs)
            tasks.append(t)
        if not tasks:
            return
        arr = (_native.OptTask * len(tasks))(*tasks)
        dev = group["params"][0].device
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _native.check(_native.lib().pfrl_rmsprop_fused_step(
            len(tasks), ctypes.cast(arr, ctypes.c_void_p), float(group["lr"]), float(group["alpha"]),
            float(group["eps"]), float(group["weight_decay"]), int(centered), stream),
            "rmsprop_fused_step")

    def step_pairs(self, pairs):
        """The RMSprop step of the given ``(parameter, finished gradient)`` pairs only, as one
        launch on the CURRENT stream (pfrl_rmsprop_step: the arithmetic of ``step()``, element by
        element).  For a caller that has some gradients early and a stream to spare -- the
        data-parallel update steps the hidden layer on the communicator's side stream, beside the
        convolution backward (GraphedUpdate) -- and marks them ``GradSource.done()`` for the
        optimizer's own launch.  False = outside what the kernel covers, nothing was done."""
        if not self.accepts_sources() or not pairs:
            return False
        group = self.param_groups[0]
        centered = bool(group["centered"])
        for p, g in pairs:
            if not (p.is_cuda and p.dtype == torch.float32 and _dense(p) and g.dtype == torch.float32
                    and g.stride() == p.stride()):
                return False
        for p, _ in pairs:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device)
                              if group.get("capturable", False) else torch.tensor(0.0))
                st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if centered:
                    st["grad_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        n = len(pairs)
        P = (ctypes.c_void_p * n)(*[p.data_ptr() for p, _ in pairs])
        G = (ctypes.c_void_p * n)(*[g.data_ptr() for _, g in pairs])
        S = (ctypes.c_void_p * n)(*[self.state[p]["square_avg"].data_ptr() for p, _ in pairs])
        A = (ctypes.c_void_p * n)(*[self.state[p]["grad_avg"].data_ptr() if centered else 0
                                    for p, _ in pairs])
        L = (ctypes.c_int64 * n)(*[p.numel() for p, _ in pairs])
        stream = ctypes.c_void_p(torch.cuda.current_stream(pairs[0][0].device).cuda_stream)
        _native.check(_native.lib().pfrl_rmsprop_step(
            n, P, G, S, A, L, float(group["lr"]), float(group["alpha"]), float(group["eps"]),
            float(group["weight_decay"]), int(centered), stream), "rmsprop_step")
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # Fusability is decided for ALL groups before anything is launched: falling back to
        # torch's step() in the middle would step the groups already updated a second time, and
        # the step happens whether or not a closure was given.
        plan = []
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            # The update is elementwise, so any dense layout works as long as the
            # parameter, its gradient and its state share it (channels_last conv
            # weights are dense permutations: the kernel walks storage order).
            ok = self._fusable(group) and all(
                p.is_cuda and p.dtype == torch.float32 and _dense(p) and not p.grad.is_sparse
                and p.grad.dtype == torch.float32 and p.grad.stride() == p.stride()
                for p in params)
            ok = ok and all(len(self.state[p]) == 0 or self.state[p]["square_avg"].stride() == p.stride()
                            for p in params)
            if not ok:
                super().step(closure=None)
                return loss
            plan.append((group, params))
        for group, params in plan:
            centered = bool(group["centered"])
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device)
                                  if group.get("capturable", False) else torch.tensor(0.0))
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if centered:
                        st["grad_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            n = len(params)
            P = (ctypes.c_void_p * n)(*[p.data_ptr() for p in params])
            G = (ctypes.c_void_p * n)(*[p.grad.data_ptr() for p in params])
            S = (ctypes.c_void_p * n)(*[self.state[p]["square_avg"].data_ptr() for p in params])
            A = (ctypes.c_void_p * n)(*[self.state[p]["grad_avg"].data_ptr() if centered else 0
                                        for p in params])
            L = (ctypes.c_int64 * n)(*[p.numel() for p in params])
            stream = ctypes.c_void_p(torch.cuda.current_stream(params[0].device).cuda_stream)
            _native.check(_native.lib().pfrl_rmsprop_step(
                n, P, G, S, A, L, float(group["lr"]), float(group["alpha"]), float(group["eps"]),
                float(group["weight_decay"]), int(centered), stream), "rmsprop_step")
        return loss


class FusedAdam(torch.optim.Adam):
    """``torch.optim.Adam`` (same constructor, same state keys ``step`` / ``exp_avg`` /
    ``exp_avg_sq``, the arithmetic of its single-tensor implementation) whose ``step()`` is
    pfrl_adam_step: one launch for all parameters of a group, step counters included
    (they live on the device, as with ``capturable=True``, so the step can be captured in a
    graph).  torch's own ``fused=True`` kernel walks 65536-element chunks, i.e. ~7 workgroups
    for the 170 k parameters of the SAC example networks: 40 us per step on MI355X against
    ~5 us here.  Configurations outside the kernel (amsgrad, maximize, non-f32, CPU, sparse)
    take torch's step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0,
                 amsgrad=False, **kw):
        kw.pop("fused", None)
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                         amsgrad=amsgrad, **kw)
    def _fusable(self, group, params, slabs=None):
        def grad_ok(p):
            if slabs and p.data_ptr() in slabs:
                return p.is_contiguous()       # (slabs hold the row-major gradient)
            return (not p.grad.is_sparse and p.grad.dtype == torch.float32
                    and p.grad.stride() == p.stride())

        return (not group["amsgrad"] and not group.get("maximize", False)
                and not group.get("differentiable", False)
                and not isinstance(group["lr"], torch.Tensor)
                and all(p.is_cuda and p.dtype == torch.float32 and _dense(p) and grad_ok(p)
                        for p in params))

    @staticmethod
    @torch.no_grad()
    def step_together(optimizers, slabs=None, soft=None, tau=0.0):
        """``step()`` of several optimizers; one launch for all of them when they are FusedAdam
        instances with one parameter group each and equal hyperparameters (the two critics of
        SAC / TD3), else one after the other.  ``slabs`` / ``soft`` / ``tau``: see ``_launch``;
        returns True when the soft updates rode in the launch (else the caller still owes them)."""
        opts = list(optimizers)
        if slabs or soft:
            return FusedAdam._step_together_ex(opts, slabs or {}, soft or {}, tau)
        same = (len(opts) > 1 and all(type(o) is FusedAdam and len(o.param_groups) == 1 for o in opts))
        if same:
            keys = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize")
            ref = opts[0].param_groups[0]
            same = all(all(o.param_groups[0].get(k) == ref.get(k) for k in keys) for o in opts)
        if not same:
            for o in opts:
                o.step()
            return
        groups = [o.param_groups[0] for o in opts]
        params = [[p for p in g["params"] if p.grad is not None] for g in groups]
        if (not all(params) or not all(o._fusable(g, ps) for o, g, ps in zip(opts, groups, params))
                or len({ps[0].device for ps in params}) != 1):
            for o in opts:
                o.step()
            return
        for o, g, ps in zip(opts, groups, params):
            if not o._prepare_state(g, ps):
                for oo in opts:
                    oo.step()
                return
        flat = [(o, p) for o, ps in zip(opts, params) for p in ps]
        opts[0]._launch(ref, [p for _, p in flat], [o.state[p] for o, p in flat])

    @staticmethod
    def _step_together_ex(opts, slabs, soft, tau):
        def has_grad(p):
            return p.grad is not None or p.data_ptr() in slabs

        ok = all(type(o) is FusedAdam and len(o.param_groups) == 1 for o in opts)
        if ok and len(opts) > 1:
            keys = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize")
            ref = opts[0].param_groups[0]
            ok = all(all(o.param_groups[0].get(k) == ref.get(k) for k in keys) for o in opts)
        groups = [o.param_groups[0] for o in opts] if ok else []
        params = [[p for p in g["params"] if has_grad(p)] for g in groups]
        ok = (ok and all(params) and len({ps[0].device for ps in params}) == 1
              and all(o._fusable(g, ps, slabs) for o, g, ps in zip(opts, groups, params))
              and all(o._prepare_state(g, ps) for o, g, ps in zip(opts, groups, params)))
        if not ok:
            for o in opts:
                FusedAdam.materialize_slabs([p for g in o.param_groups for p in g["params"]], slabs)
                o.step()
            return False
        flat = [(o, p) for o, ps in zip(opts, params) for p in ps]
        stepped = {p.data_ptr(): p for _, p in flat}
        soft_ok = bool(soft) and set(soft) <= set(stepped) and all(
            t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            and stepped[k].is_contiguous() and t.shape == stepped[k].shape for k, t in soft.items())
        opts[0]._launch(groups[0], [p for _, p in flat], [o.state[p] for o, p in flat], slabs,
                        soft if soft_ok else None, tau)
        return soft_ok

    def _prepare_state(self, group, params):
        """Device-side step counters (as torch's capturable path keeps them, which is also what
        its own step() needs should a later call fall outside the kernel) and moment buffers;
        False if an existing state does not have the parameters' layout."""
        group["capturable"] = True
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not st["step"].is_cuda:
                # state loaded from a checkpoint of a host-step optimizer
                st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
            if st["exp_avg"].stride() != p.stride() or st["exp_avg_sq"].stride() != p.stride():
                return False
        return True

    def _launch(self, group, params, states, slabs=None, soft=None, tau=0.0):
        """``slabs``: {parameter data_ptr: (partial slabs, stride, n)} -- gradients the backward
        pass left as split-K slabs (``nn.mfma_linear.slab_sink``), summed inside the launch;
        ``soft``: {parameter data_ptr: target tensor} soft-updated with ``tau`` from the new values."""
        dev = params[0].device
        tickets = self.__dict__.setdefault("_ticket", {})
        if dev not in tickets:
            tickets[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        n = len(params)
        V = ctypes.c_void_p
        slabs = slabs or {}
        soft = soft or {}
        src = [slabs.get(p.data_ptr()) for p in params]
        P = (V * n)(*[p.data_ptr() for p in params])
        G = (V * n)(*[(s[0].data_ptr() if s is not None else p.grad.data_ptr())
                      for p, s in zip(params, src)])
        M = (V * n)(*[st["exp_avg"].data_ptr() for st in states])
        S = (V * n)(*[st["exp_avg_sq"].data_ptr() for st in states])
        T = (V * n)(*[st["step"].data_ptr() for st in states])
        L = (ctypes.c_int64 * n)(*[p.numel() for p in params])
        b1, b2 = group["betas"]
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if not slabs and not soft:
            _native.check(_native.lib().pfrl_adam_step(
                n, P, G, M, S, T, L, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                float(group["weight_decay"]), V(tickets[dev].data_ptr()), stream), "adam_step")
            return
        SS = (ctypes.c_int64 * n)(*[(int(s[1]) if s is not None else 0) for s in src])
        NS = (ctypes.c_int32 * n)(*[(int(s[2]) if s is not None else 1) for s in src])
        D = (V * n)(*[(soft[p.data_ptr()].data_ptr() if p.data_ptr() in soft else 0) for p in params])
        _native.check(_native.lib().pfrl_adam_step_ex(
            n, P, G, SS, NS, M, S, T, D, float(tau), L, float(group["lr"]), float(b1), float(b2),
            float(group["eps"]), float(group["weight_decay"]), V(tickets[dev].data_ptr()), stream),
            "adam_step_ex")

    @staticmethod
    def materialize_slabs(params, slabs):
        """Fallback for a step the kernel does not cover: sum the slabs into ``.grad``."""
        for p in params:
            s = slabs.get(p.data_ptr()) if slabs else None
            if s is not None and p.grad is None:
                part, stride, n = s
                flat = torch.as_strided(part, (int(n), p.numel()), (int(stride), 1))
                p.grad = flat.sum(0).view_as(p)

    @torch.no_grad()
    def step(self, closure=None, slabs=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # decide for ALL groups before launching anything: a mix of fusable and non-fusable
        # groups takes torch's step as a whole (launching some groups here and then calling
        # super().step() would step those groups twice)
        work = []
        for group in self.param_groups:
            params = [p for p in group["params"]
                      if p.grad is not None or (slabs and p.data_ptr() in slabs)]
            if not params:
                continue
            if not self._fusable(group, params, slabs) or not self._prepare_state(group, params):
                self.materialize_slabs([p for g in self.param_groups for p in g["params"]], slabs)
                super().step(closure=None)
                return loss
            work.append((group, params))
        for group, params in work:
            self._launch(group, params, [self.state[p] for p in params], slabs)
        return loss


class RMSpropEpsInsideSqrt(torch.optim.RMSprop):
    """``torch.optim.RMSprop`` whose denominator is ``sqrt(v + eps)`` instead of ``sqrt(v) + eps``
    -- the A3C / A2C papers' form, used by examples/atari/train_a2c_ale.py (reference
    pfrl/optimizers/rmsprop_eps_inside_sqrt.py:5-63).  Constructor and state keys (``step``,
    ``square_avg``, ``grad_avg``, ``momentum_buffer``) are the reference's; the arithmetic runs
    as a handful of multi-tensor (foreach) launches per parameter group rather than per tensor."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if any(p.grad.is_sparse for p in params):
                raise RuntimeError("RMSprop does not support sparse gradients")
            alpha, eps, lr = group["alpha"], group["eps"], group["lr"]
            momentum, centered = group["momentum"], group["centered"]
            for p in params:
                self._init_state(p, group)
                self.state[p]["step"] += 1
            grads = [p.grad for p in params]
            if group["weight_decay"] != 0:
                grads = torch._foreach_add(grads, params, alpha=group["weight_decay"])
            square_avg = [self.state[p]["square_avg"] for p in params]
            torch._foreach_mul_(square_avg, alpha)
            torch._foreach_addcmul_(square_avg, grads, grads, value=1 - alpha)
            if centered:
                grad_avg = [self.state[p]["grad_avg"] for p in params]
                torch._foreach_mul_(grad_avg, alpha)
                torch._foreach_add_(grad_avg, grads, alpha=1 - alpha)
                denom = torch._foreach_addcmul(square_avg, grad_avg, grad_avg, value=-1)
                torch._foreach_add_(denom, eps)
            else:
                denom = torch._foreach_add(square_avg, eps)
            torch._foreach_sqrt_(denom)
            if momentum > 0:
                bufs = [self.state[p]["momentum_buffer"] for p in params]
                torch._foreach_mul_(bufs, momentum)
                torch._foreach_addcdiv_(bufs, grads, denom)
                torch._foreach_add_(params, bufs, alpha=-lr)
            else:
                torch._foreach_addcdiv_(params, grads, denom, value=-lr)
        return loss

    def _init_state(self, p, group):
        state = self.state[p]
        if len(state) == 0:
            state["step"] = 0
            state["square_avg"] = torch.zeros_like(p)
            if group["momentum"] > 0:
                state["momentum_buffer"] = torch.zeros_like(p)
            if group["centered"]:
                state["grad_avg"] = torch.zeros_like(p)


class SharedRMSpropEpsInsideSqrt(RMSpropEpsInsideSqrt):
    """The same with the state allocated at construction, so that it exists before the
    parameters are shared between processes (reference :66-82)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for group in self.param_groups:
            for p in group["params"]:
                self._init_state(p, group)
from pfrl_amd.optimizers import rmsprop_eps_inside_sqrt  # NOQA,E402  (reference module path)
