"""``nn.Linear`` (+ ReLU) of the MLP agents as gfx950 MFMA kernels at minibatch size.

The MLPs of the replay actor-critic agents (256-256 policy and twin Q of
``examples/mujoco/reproduction/soft_actor_critic/train_soft_actor_critic.py:172-243``, the
400-300 nets of TD3 / DDPG) are evaluated ~26 times per update at B = 100-256
(``pfrl/agents/soft_actor_critic.py:213-330``).  Measured on MI355X inside a captured graph
(``tools/linear_check.py``), the library picks one-workgroup tiles for these shapes: an
``F.linear`` with a 256 x 256 output takes 63-96 us, and so do the 256 x 256 gradients.  The
implicit-GEMM engine of ``csrc/qnet.hip`` runs the same layers (the 1 x 1 case) in 5-7 us with
the bias and ReLU in its epilogue and the ReLU mask folded into both gradient kernels.

``accelerate_mlp(model)`` changes no parameter, name or state_dict key: plain ``nn.Linear``
children become ``_LinearSlot`` (an ``nn.Linear`` subclass around the same tensors) and plain
``nn.Sequential`` containers become ``_MlpSequential`` (same children) so that a ``Linear``
followed by ``nn.ReLU`` is one node.  Inputs outside what the kernels cover (CPU tensors,
other dtypes, batches above ``PFRL_MFMA_LINEAR_MAX_BATCH``) take ``F.linear`` as before.

Numerics: f32 in, f32 accumulate (exact fmaf chains); only the summation order differs from
hipBLASLt's.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from pfrl_amd import _native
from pfrl_amd._native import check
from pfrl_amd.nn import mfma_trunk as _t
from pfrl_amd.nn.mfma_trunk import _ceil_div, _p, _stream

_ENABLED = os.environ.get("PFRL_MFMA_LINEAR", "1") != "0"
_MAX_BATCH = int(os.environ.get("PFRL_MFMA_LINEAR_MAX_BATCH", "1024"))
# output tiles (16 x 32) from which a layer is launched without split-K
_DIRECT_TILES = int(os.environ.get("PFRL_MFMA_LINEAR_DIRECT_TILES", "96"))
MIN_OUT = 17   # narrower layers are the narrow-head kernels' (mfma_trunk._SmallLinear)


# While a ``slab_sink()`` is open, weight / bias gradients that the backward kernels leave as
# split-K slabs are NOT folded (pfrl_splitk_reduce) and NOT returned: the slabs are recorded as
# {parameter data_ptr: (slabs, stride, n)} for an optimizer that sums them inside its own launch
# (FusedAdam.step(slabs=...), pfrl_adam_step_ex).  The caller owns the contract: ``.grad`` of those
# parameters stays None, so nothing between backward and the step may read it.
_SLAB_SINK = None


class slab_sink:
    def __enter__(self):
        global _SLAB_SINK
        self._prev = _SLAB_SINK
        _SLAB_SINK = self.slabs = {}
        return self.slabs

    def __exit__(self, *exc):
        global _SLAB_SINK
        _SLAB_SINK = self._prev
        return False


def _fold_or_sink(part, stride, splits, w_ptr, b_ptr, dw, db, nW, Fo):
    """Returns (dw, db) after folding the slabs -- or (None, None) with the slabs in the sink."""
    sink = _SLAB_SINK
    if sink is not None and b_ptr is not None:
        sink[w_ptr] = (part, stride, splits)
        sink[b_ptr] = (part[nW:], stride, splits)
        return None, None
    _t._reduce([(part, dw, None, stride, nW, splits, 4, 0),
                (part[nW:], db, None, stride, Fo, splits, 4, 0)])
    return dw, db


def _fwd_splits(M, Fo, K):
    nch = _ceil_div(K, 32)
    tiles16 = _ceil_div(M, 16) * _ceil_div(Fo, 32)
    if tiles16 >= _DIRECT_TILES or Fo % 4 != 0:   # (the fold kernel moves float4 columns)
        return 1
    want = min(max(448 // tiles16, _ceil_div(nch, 16), 1), nch)
    cps = _ceil_div(nch, want)
    return _ceil_div(nch, cps)


def _bwd_kernels_cover(M, K, Fo):
    """Both gradient kernels need a whole number of 32-chunks in their reductions."""
    return K % 32 == 0 and Fo % 32 == 0


class _Linear(torch.autograd.Function):
    """y = act(x w^T + b), act = ReLU or identity, as one autograd node."""

    @staticmethod
    def forward(ctx, x, w, b, relu, splits=None):
        M, K = x.shape
        Fo = w.shape[0]
        x = x.contiguous()
        lib = _native.lib()
        y = torch.empty((M, Fo), dtype=torch.float32, device=x.device)
        splits = _fwd_splits(M, Fo, K) if splits is None else int(splits)
        if splits == 1:
            check(lib.pfrl_linear_fwd(_p(x), _p(w), _p(b), _p(y), M, K, Fo, int(relu), 1, _stream()),
                  "linear_fwd")
        else:
            part = torch.empty((splits, M, Fo), dtype=torch.float32, device=x.device)
            check(lib.pfrl_linear_fwd(_p(x), _p(w), None, _p(part), M, K, Fo, 0, splits, _stream()),
                  "linear_fwd_splitk")
            _t._reduce([(part, y, b, M * Fo, M * Fo, splits, Fo, int(relu))])
        ctx.relu = bool(relu)
        ctx.b_ptr = b.data_ptr() if b is not None else None
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        Fo = w.shape[0]
        need_dx = ctx.needs_input_grad[0]
        dev = x.device
        if (not _bwd_kernels_cover(M, K, Fo) and not ctx.relu and Fo <= 64 and Fo % 16 != 0
                and _ceil_div(M, 16) * 16 * (_ceil_div(Fo, 16) * 16 + 1) * 4 <= 65536
                and ctx.needs_input_grad[1]):
            # the 2 * action_size policy head (Linear(256, 34) of SAC): dx, dw and db from the
            # narrow-head launch (library route below: two products and a column sum)
            dyc = dy.contiguous()
            dx = torch.empty_like(x) if need_dx else None
            dw = torch.empty_like(w)
            db = torch.empty(Fo, dtype=torch.float32, device=dev)
            check(_native.lib().pfrl_linear_small_bwd(_p(dyc), _p(x), _p(w), _p(dx), _p(dw), _p(db), M, K,
                                                      Fo, _stream()), "linear_small_bwd")
            return dx, dw, db, None, None
        if not _bwd_kernels_cover(M, K, Fo):
            # ragged layers (first layer of an MLP, 2 * action_size heads)
            need_w = ctx.needs_input_grad[1]
            dx = dw = db = None
            g = None
            if need_dx or not (need_w and Fo % 16 == 0):
                # library products: their outputs are not the 256 x 256 shapes that the
                # library's heuristics mishandle
                g = torch.ops.aten.threshold_backward(dy, y, 0.0) if ctx.relu else dy
            if need_dx:
                dx = g @ w
            if need_w and Fo % 16 == 0:
                # any in_features: the TAIL weight-gradient kernel, ReLU mask and db folded in
                dyc = dy.contiguous()
                dw = torch.empty_like(w)
                db = torch.empty(Fo, dtype=torch.float32, device=dev)
                splits = _t._wgrad_splits(M, Fo, _ceil_div(K, 32) * 32)
                nW = w.numel()
                stride = nW + Fo
                if splits == 1:
                    pw, pb, st = dw, db, 0
                else:
                    part = torch.empty(splits * stride, dtype=torch.float32, device=dev)
                    pw, pb, st = part, part[nW:], stride
                check(_native.lib().pfrl_linear_bwd_weight(_p(dyc), _p(y), _p(x), _p(pw), _p(pb), st, st,
                                                           M, K, Fo, splits, _stream()),
                      "linear_bwd_weight")
                if splits > 1:
                    dw, db = _fold_or_sink(part, stride, splits, w.data_ptr(), ctx.b_ptr, dw, db, nW, Fo)
            elif need_w:
                dw = g.t() @ x
                db = g.sum(0) if ctx.needs_input_grad[2] else None
            return dx, dw, db, None, None
        dy = dy.contiguous()
        lib = _native.lib()
        if not ctx.needs_input_grad[1]:
            # frozen weights (e.g. the Q-networks under the SAC policy loss): input gradient only
            dx = None
            if need_dx:
                dx = torch.empty_like(x)
                check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dy), _p(y), _p(w), None, _p(dx), M, 1, 1, K, Fo,
                                                    1, 1, 1, 0, 0, _stream()), "linear_bwd_data")
            return dx, None, None, None, None
        dw = torch.empty_like(w)
        db = torch.empty(Fo, dtype=torch.float32, device=dev)
        splits = _t._wgrad_splits(M, Fo, K)
        nW = w.numel()
        stride = nW + Fo
        if splits == 1:
            pw, pb, st = dw, db, 0
        else:
            part = torch.empty(splits * stride, dtype=torch.float32, device=dev)
            pw, pb, st = part, part[nW:], stride
        dx = None
        if need_dx:
            dx = torch.empty_like(x)
            if _t._fused_bwd_ok(M, 1, 1, K, 1):
                check(lib.pfrl_conv2d_nhwc_bwd(_p(dy), _p(y), _p(w), None, _p(x), _p(dx), _p(pw), _p(pb),
                                               st, st, M, 1, 1, K, Fo, 1, 1, 1, 0, 0, splits, _stream()),
                      "linear_bwd")
            else:
                check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dy), _p(y), _p(w), None, _p(dx), M, 1, 1, K, Fo,
                                                    1, 1, 1, 0, 0, _stream()), "linear_bwd_data")
                check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dy), _p(y), _p(x), _p(pw), _p(pb), st, st, M, 1,
                                                      1, K, Fo, 1, 1, 1, splits, _stream()),
                      "linear_bwd_weight")
        else:
            check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dy), _p(y), _p(x), _p(pw), _p(pb), st, st, M, 1, 1,
                                                  K, Fo, 1, 1, 1, splits, _stream()),
                  "linear_bwd_weight")
        if splits > 1:
            dw, db = _fold_or_sink(part, stride, splits, w.data_ptr(), ctx.b_ptr, dw, db, nW, Fo)
        return dx, dw, db, None, None


def _reduce_noisy(part, y, mu_b, sigma_b, r_out, n, splits, ncol, relu):
    import ctypes

    V = ctypes.c_void_p
    one = lambda t: (V * 1)(t.data_ptr())    # noqa: E731
    check(_native.lib().pfrl_splitk_reduce_noisy(
        1, one(part), one(y), one(mu_b), one(sigma_b), one(r_out), (ctypes.c_int64 * 1)(n),
        (ctypes.c_int32 * 1)(n), (ctypes.c_int32 * 1)(splits), (ctypes.c_int32 * 1)(ncol),
        (ctypes.c_int32 * 1)(int(relu)), _stream()), "splitk_reduce_noisy")


class _Ctx:
    """What ``_Linear.backward`` reads from its context."""

    def __init__(self, saved, needs, relu):
        self.saved_tensors, self.needs_input_grad, self.relu, self.b_ptr = saved, needs, relu, None


class _NoisyLinear(torch.autograd.Function):
    """``act(x W^T + b)`` of a factorised NoisyNet layer as one node that never writes W in its
    forward pass (pfrl_linear_noisy_fwd: the perturbed weights are formed inside the kernel's
    operand loader; bit-identical to noisy_weights + _Linear).  The no-grad passes of an update
    (target network, Double-DQN action selection, acting) need nothing else; a pass that is
    differentiated builds W once in its BACKWARD (where the launch is off the path the next
    minibatch waits for) and continues as _Linear.backward / pfrl_noisy_weights_bwd do."""

    @staticmethod
    def forward(ctx, x, mu_w, sigma_w, mu_b, sigma_b, r, relu):
        M, K = x.shape
        Fo = mu_w.shape[0]
        x = x.contiguous()
        lib = _native.lib()
        y = torch.empty((M, Fo), dtype=torch.float32, device=x.device)
        splits = noisy_fwd_splits(M, Fo, K)
        if splits == 1:
            check(lib.pfrl_linear_noisy_fwd(_p(x), _p(mu_w), _p(sigma_w), _p(mu_b), _p(sigma_b), _p(r),
                                            _p(y), M, K, Fo, int(relu), 1, _stream()), "linear_noisy_fwd")
        else:
            part = torch.empty((splits, M, Fo), dtype=torch.float32, device=x.device)
            check(lib.pfrl_linear_noisy_fwd(_p(x), _p(mu_w), _p(sigma_w), None, None, _p(r), _p(part), M,
                                            K, Fo, 0, splits, _stream()), "linear_noisy_fwd_splitk")
            _reduce_noisy(part, y, mu_b, sigma_b, r[K:], M * Fo, splits, Fo, relu)
        ctx.relu = bool(relu)
        ctx.save_for_backward(x, mu_w, sigma_w, r, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mu_w, sigma_w, r, y = ctx.saved_tensors
        Fo, K = mu_w.shape
        lib = _native.lib()
        w = torch.empty_like(mu_w)
        check(lib.pfrl_noisy_weights_fwd(_p(mu_w), _p(sigma_w), None, None, _p(r), _p(w), None, Fo, K,
                                         _stream()), "noisy_weights_fwd")
        need_w = any(ctx.needs_input_grad[1:5])
        dx, dw, db, _, _ = _Linear.backward(
            _Ctx((x, w, y), (ctx.needs_input_grad[0], need_w, need_w, False), ctx.relu), dy)
        if not need_w:
            return dx, None, None, None, None, None, None
        dw, db = dw.contiguous(), db.contiguous()
        g_sw, g_sb = torch.empty_like(dw), torch.empty_like(db)
        check(lib.pfrl_noisy_weights_bwd(_p(dw), _p(db), _p(r), _p(g_sw), _p(g_sb), Fo, K, _stream()),
              "noisy_weights_bwd")
        return dx, dw, g_sw, db, g_sb, None, None


class _NoisyLinearPair(torch.autograd.Function):
    """Two NoisyNet layers on the two halves of one activation tensor -- the advantage and value
    streams of the distributional dueling head -- as ONE forward launch reading the halves in place
    (pfrl_linear_noisy_fwd_pair).  Backward: what the two single-layer nodes and the split of ``h``
    do (contiguous halves, _NoisyLinear.backward each, the two input gradients side by side)."""

    @staticmethod
    def forward(ctx, h, mu_w0, sigma_w0, mu_b0, sigma_b0, r0, mu_w1, sigma_w1, mu_b1, sigma_b1, r1):
        import ctypes

        M, K2 = h.shape
        K = K2 // 2
        h = h.contiguous()
        N0, N1 = mu_w0.shape[0], mu_w1.shape[0]
        y0 = torch.empty((M, N0), dtype=torch.float32, device=h.device)
        y1 = torch.empty((M, N1), dtype=torch.float32, device=h.device)
        V = ctypes.c_void_p
        two = lambda a, b: (V * 2)(a, b)                                  # noqa: E731
        check(_native.lib().pfrl_linear_noisy_fwd_pair(
            two(h.data_ptr(), h.data_ptr() + 4 * K), K2, two(_p(mu_w0), _p(mu_w1)),
            two(_p(sigma_w0), _p(sigma_w1)), two(_p(mu_b0), _p(mu_b1)), two(_p(sigma_b0), _p(sigma_b1)),
            two(_p(r0), _p(r1)), two(_p(y0), _p(y1)), M, K, (ctypes.c_int32 * 2)(N0, N1), 0, _stream()),
            "linear_noisy_fwd_pair")
        ctx.save_for_backward(h, mu_w0, sigma_w0, r0, mu_w1, sigma_w1, r1)
        return y0, y1

    @staticmethod
    def backward(ctx, dy0, dy1):
        h, mu_w0, sigma_w0, r0, mu_w1, sigma_w1, r1 = ctx.saved_tensors
        M, K2 = h.shape
        K = K2 // 2
        halves = h.view(M, 2, K).transpose(0, 1).contiguous()
        need = ctx.needs_input_grad
        outs, dxs = [], []
        for x, mu_w, sigma_w, r, dy, ni in ((halves[0], mu_w0, sigma_w0, r0, dy0, 1),
                                            (halves[1], mu_w1, sigma_w1, r1, dy1, 6)):
            if dy is None:
                dy = torch.zeros((M, mu_w.shape[0]), dtype=torch.float32, device=h.device)
            sub = _Ctx((x, mu_w, sigma_w, r, None), (need[0],) + tuple(need[ni:ni + 4]) + (False, False),
                       False)
            g = _NoisyLinear.backward(sub, dy)
            dxs.append(g[0])
            outs.append(g[1:5])
        dh = torch.cat(dxs, dim=1) if need[0] else None
        return (dh,) + tuple(outs[0]) + (None,) + tuple(outs[1]) + (None,)


def noisy_pair_supported(h, a, v):
    """Both streams are factorised-noise layers on the halves of ``h`` and fall into the
    narrow-output tile program of the forward kernel (what pfrl_linear_noisy_fwd_pair covers)."""
    if not (h.is_cuda and h.dim() == 2 and h.shape[1] % 2 == 0 and a.hasbias and v.hasbias):
        return False
    K = h.shape[1] // 2
    ma, mv = a.mu.weight, v.mu.weight
    if ma.shape[1] != K or mv.shape[1] != K or K % 32 != 0:
        return False
    M = h.shape[0]
    narrow = all(w.shape[0] % 32 != 0 and _ceil_div(M, 64) * _ceil_div(w.shape[0], 16) < 512
                 and _fwd_splits(M, w.shape[0], K) == 1 for w in (ma, mv))
    return (narrow and os.environ.get("PFRL_NOISY_PAIR", "1") != "0"
            and noisy_supported(h[:, :K], ma, a.sigma.weight, a.mu.bias, a.sigma.bias)
            and noisy_supported(h[:, :K], mv, v.sigma.weight, v.mu.bias, v.sigma.bias))


def noisy_fwd_splits(M, Fo, K):
    """Split-K of a NoisyNet layer's forward: twice the plain layer's (the loader streams mu AND
    sigma: 25.7 MB for the 3136 x 1024 layer of the Rainbow head, and the launch is bound by how
    many workgroups have loads in flight -- 7 -> 14 splits: Rainbow +4 %, 28: +0.6 % more).  Used by
    both noisy paths (in-loader and materialised), so that they stay bit-identical."""
    s = _fwd_splits(M, Fo, K)
    if s == 1:
        return 1
    if os.environ.get("PFRL_NOISY_SPLITS"):      # (A/B experiments)
        return max(1, min(_ceil_div(K, 32), int(os.environ["PFRL_NOISY_SPLITS"])))
    nch = _ceil_div(K, 32)
    cps = _ceil_div(nch, min(nch, 2 * s))
    return _ceil_div(nch, cps)


def noisy_supported(x, mu_w, sigma_w, mu_b, sigma_b):
    """The in-kernel NoisyNet forward covers minibatch- and acting-sized batches of layers whose
    in_features are a multiple of 32 (PFRL_NOISY_IN_LOADER=0: materialised weights as before)."""
    return (_ENABLED and os.environ.get("PFRL_NOISY_IN_LOADER", "1") != "0" and x.is_cuda
            and x.dim() == 2 and x.dtype == torch.float32 and 0 < x.shape[0] <= _MAX_BATCH
            and mu_b is not None and sigma_b is not None and mu_w.shape[0] >= MIN_OUT
            and mu_w.shape[1] % 32 == 0 and mu_w.is_contiguous() and sigma_w.is_contiguous()
            and mu_w.dtype == torch.float32 and _native.available())


def supported(layer, x):
    return (_ENABLED and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and layer.bias is not None and layer.out_features >= MIN_OUT
            and layer.weight.dtype == torch.float32 and layer.weight.is_contiguous()
            and 0 < x.shape[0] <= _MAX_BATCH and _native.available())


def supported_tensors(x, weight, bias):
    """``supported`` for a weight / bias pair that is not an ``nn.Linear`` (noisy weights)."""
    return (_ENABLED and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and bias is not None and weight.shape[0] >= MIN_OUT and weight.dtype == torch.float32
            and weight.is_contiguous() and 0 < x.shape[0] <= _MAX_BATCH and _native.available())


class _LinearSlot(nn.Linear):
    """An ``nn.Linear`` (same class family, parameter names and state_dict) around the
    tensors of an existing layer; forward takes the MFMA kernels where ``supported``."""

    def __init__(self, linear):
        nn.Module.__init__(self)
        self.__dict__.update({k: v for k, v in linear.__dict__.items()
                              if k not in ("_parameters", "_buffers", "_modules")})
        self._parameters = linear._parameters
        self._buffers = linear._buffers
        self._modules = linear._modules

    def forward(self, x, relu=False):
        if supported(self, x):
            return _Linear.apply(x, self.weight, self.bias, relu)
        y = nn.Linear.forward(self, x)
        return F.relu(y) if relu else y


class _MlpSequential(nn.Sequential):
    """An ``nn.Sequential`` (same children, indices, state_dict) that runs a ``_LinearSlot``
    followed by a plain ``nn.ReLU`` as one node."""

    def forward(self, x):
        mods = list(self._modules.values())
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            if (isinstance(m, _LinearSlot) and i + 1 < n and type(mods[i + 1]) is nn.ReLU
                    and torch.is_tensor(x) and supported(m, x)):
                x = m(x, relu=True)
                i += 2
                continue
            x = m(x)
            i += 1
        return x


def accelerate_mlp(model):
    """Route the ``nn.Linear`` layers of ``model`` (plain class, with bias, out_features >= 17)
    through the MFMA kernels and fuse ``Linear, ReLU`` neighbours of plain ``nn.Sequential``
    containers; narrower layers get the narrow-head kernels (``accelerate_heads``).  Returns
    ``model`` (modified in place; parameters, names and state_dict keys unchanged)."""
    if model is None:
        return model
    for parent in list(model.modules()):
        for name, child in list(parent._modules.items()):
            if type(child) is nn.Linear and child.bias is not None and child.out_features >= MIN_OUT:
                parent._modules[name] = _LinearSlot(child)
    for seq in [m for m in model.modules() if type(m) is nn.Sequential]:
        if any(isinstance(c, _LinearSlot) for c in seq._modules.values()):
            seq.__class__ = _MlpSequential
    _t.accelerate_heads(model)
    return model
