"""Conv trunk + hidden linear layer of the example Q-networks as hand-written gfx950
kernels (``csrc/qnet.hip``): ``activation(layer(h))`` of ``pfrl/nn/atari_cnn.py:40-47``
and of the ``nn.Sequential`` in ``examples/atari/train_ppo_ale.py:247-264``, forward and
backward, for ReLU networks whose convolutions keep channels_last weights.

Why: at the minibatch of the replay agents (B = 32, ``pfrl/agents/dqn.py:316-365``, 64
dependent updates per batched env step) the MIOpen / hipBLASLt route is ~33 launches of
5-17 us per update, each with <1 us of work.  Here one update's network part is 5
forward and 8 backward launches: implicit-GEMM f32 MFMA kernels with bias + ReLU (forward)
and ReLU mask (backward) in the epilogues, split-K partials folded by one multi-tensor
launch, no zero-fill helpers, no layout copies.

Numerics: f32 in, f32 accumulate (``v_mfma_f32_16x16x4_f32`` is an exact fmaf chain);
only the summation order differs from MIOpen's.

The whole trunk is ONE autograd node, so every intermediate layout is private: the last
convolution's output is written planar (NCHW) because the linear layer's weight columns
are in that order, and the linear layer's input gradient is written back as NHWC rows for
the convolution backward.  There is no fallback inside: ``supported()`` decides up
front, and unsupported shapes take the stock PyTorch route.
"""
import contextlib
import ctypes
import os

import torch
import torch.nn as nn

from pfrl_amd import _native
from pfrl_amd._native import check

_ENABLED = os.environ.get("PFRL_MFMA_TRUNK", "1") != "0"
# batches above this keep the library kernels (0 = no limit)
_MAX_BATCH = int(os.environ.get("PFRL_MFMA_TRUNK_MAX_BATCH", "0"))
SMALL_LINEAR_MAX_OUT = 16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # (torch.cuda.current_stream() builds a Stream object through several Python layers, ~7 us a
    # call and a dozen calls per update; the raw handle of the current device's current stream
    # is one C call)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ceil_div(a, b):
    return -(-a // b)


def _is_relu(act):
    import torch.nn.functional as F

    return act is F.relu or act is torch.relu or isinstance(act, nn.ReLU)


class ConvSpec:
    """Geometry of one convolution of the trunk (forward view)."""

    __slots__ = ("C", "Cout", "R", "S", "ST", "H", "W", "OH", "OW")

    def __init__(self, conv, H, W):
        self.C, self.Cout = conv.in_channels, conv.out_channels
        self.R, self.S = conv.kernel_size
        self.ST = conv.stride[0]
        self.H, self.W = H, W
        self.OH = (H - self.R) // self.ST + 1
        self.OW = (W - self.S) // self.ST + 1


def _conv_ok(conv):
    return (isinstance(conv, nn.Conv2d) and conv.bias is not None and conv.groups == 1
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.padding_mode == "zeros"
            and conv.stride[0] == conv.stride[1] and conv.weight.dtype == torch.float32
            and conv.weight.is_cuda
            and conv.weight.is_contiguous(memory_format=torch.channels_last)
            and (conv.kernel_size[1] * conv.in_channels) % 32 == 0 and conv.in_channels % 4 == 0
            and conv.out_channels % 16 == 0)


def u8_first_layer_shape_ok(conv, M, H, W, divisor):
    """The u8 tile programs of pfrl_conv2d_u8nhwc4_fwd / _bwd_weight (csrc/qnet.hip) cover
    ``conv`` on M images of H x W u8 NHWC4 pixels, and ``divisor`` divides every byte value with
    IEEE rounding in the loader's three operations."""
    from pfrl_amd import ops

    if not (_U8_FIRST and _conv_ok(conv) and conv.in_channels == 4
            and conv.out_channels % 32 == 0 and conv.out_channels % 64 != 0
            and H >= conv.kernel_size[0] and W >= conv.kernel_size[1]):
        return False
    st = conv.stride[0]
    M = max(M, _PLAN_BATCH)
    rows = M * ((H - conv.kernel_size[0]) // st + 1) * ((W - conv.kernel_size[1]) // st + 1)
    return (_ceil_div(rows, 32) * (conv.out_channels // 32) >= 384
            and ops.u8_division_exact(divisor))


def u8_first_layer_ok(conv, px):
    """The first convolution can read ``px`` (ops.U8Pixels) itself."""
    d = px.data
    return (d.is_cuda and d.dtype == torch.uint8 and d.dim() == 4 and d.shape[3] == 4
            and d.is_contiguous() and d.data_ptr() % 16 == 0
            and u8_first_layer_shape_ok(conv, d.shape[0], d.shape[1], d.shape[2], px.divisor))


_U8_FIRST = os.environ.get("PFRL_U8_CONV1", "1") != "0"


def plan_for(convs, linear, x):
    """ConvSpec list if (convs..., flatten, linear) on input ``x`` [N, C, H, W] (or the
    ops.U8Pixels standing for one) is inside what the kernels cover, else None."""
    from pfrl_amd.ops import U8Pixels

    if isinstance(x, U8Pixels):
        if not (_ENABLED and _native.available() and convs and u8_first_layer_ok(convs[0], x)):
            return None
    elif not (_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
              and not x.requires_grad and _native.available()):
        return None
    if _MAX_BATCH and x.shape[0] > _MAX_BATCH:
        return None
    # linear is None: the convolutions alone (networks whose layer after the flatten is not a
    # plain nn.Linear + ReLU, e.g. the NoisyNet streams of the Rainbow head)
    if linear is not None and not (
            isinstance(linear, nn.Linear) and linear.bias is not None
            and linear.weight.dtype == torch.float32 and linear.weight.is_contiguous()
            and linear.in_features % 32 == 0 and linear.out_features % 32 == 0):
        return None
    H, W = x.shape[2], x.shape[3]
    specs = []
    for i, conv in enumerate(convs):
        if not _conv_ok(conv) or H < conv.kernel_size[0] or W < conv.kernel_size[1]:
            return None
        sp = ConvSpec(conv, H, W)
        if i == 0:
            if sp.C != x.shape[1]:
                return None
        else:
            # the backward-data kernel of layers that have a layer below
            if not (sp.Cout % 32 == 0 and sp.C % 16 == 0 and sp.R % sp.ST == 0
                    and sp.S % sp.ST == 0 and sp.H % sp.ST == 0 and sp.W % sp.ST == 0
                    and sp.C == specs[-1].Cout):
                return None
        specs.append(sp)
        H, W = sp.OH, sp.OW
    last = specs[-1]
    if linear is not None and (last.Cout * last.OH * last.OW != linear.in_features
                               or linear.in_features % 16):
        return None
    if x.shape[0] * max(s.OH * s.OW * max(s.Cout, s.C) for s in specs) >= 2 ** 31 // 8:
        return None
    return specs


# Batch size the FORWARD launches are planned for (0: the batch of the call): tile programs, the
# split-K of the linear layer and the NHWC / planar route are then chosen as for a batch of that
# many images, so that a few rows evaluated on their own come out bit for bit as they would inside
# the large batch (no tile program mixes rows).  See csrc/qnet.hip::pfrl_qnet_plan_images and
# PPO._next_values_exact.  Forward passes without gradients only.
_PLAN_BATCH = 0


@contextlib.contextmanager
def plan_batch(images):
    global _PLAN_BATCH
    prev = _PLAN_BATCH
    _PLAN_BATCH = int(images)
    check(_native.lib().pfrl_qnet_plan_images(_PLAN_BATCH), "qnet_plan_images")
    try:
        yield
    finally:
        _PLAN_BATCH = prev
        check(_native.lib().pfrl_qnet_plan_images(prev), "qnet_plan_images")


def _fwd_splits(M, F, K):
    M = max(M, _PLAN_BATCH)
    nch = K // 32
    tiles16 = _ceil_div(M, 16) * _ceil_div(F, 32)
    if tiles16 >= 1024:
        return 1
    want = max(448 // tiles16, _ceil_div(nch, 16), 1)
    want = min(want, nch)
    cps = _ceil_div(nch, want)
    return _ceil_div(nch, cps)


def _wgrad_tile(M, Cout, K):
    """(rows, columns) of the weight-gradient tile program pfrl_conv2d_nhwc_bwd_weight picks
    (csrc/qnet.hip; keep in step)."""
    if M >= 16384 and Cout % 32 == 0:
        if Cout % 64 == 0 and K % 128 == 0 and M >= 262144:
            return 64, 128
        if Cout % 64 == 0 and K % 64 == 0:
            return 64, 64
        if K % 256 == 0:
            return 32, 256
        if K % 128 == 0:
            return 32, 128
    return (32, 32) if Cout % 32 == 0 else (16, 32)


def _wgrad_splits(M, Cout, K):
    nch = _ceil_div(M, 32)
    bi, bj = _wgrad_tile(M, Cout, K)
    if bi * bj > 32 * 32:
        # large tile programs (rollout / update size): what the kernel wants is ~4 000 workgroups
        # (16 per CU) of at least 16 chunks each -- measured (profiles/r04_wgrad_splits.txt): more,
        # shorter walks beat fewer, longer ones by 15-35 %, and beyond that nothing moves -- while
        # every split is a slab of the whole output that the fold launch reads back: 4 096 splits
        # for the first convolution (one tile, 33 KB slabs), ~10 for the linear layer (392 tiles,
        # 6.4 MB slabs: 32 splits cost 50 us of fold for 14 us of kernel)
        tiles = _ceil_div(Cout, bi) * _ceil_div(K, bj)
        want = min(max(_ceil_div(4096, tiles), 1), max(nch // 16, 1), _WGRAD_MAX_SPLITS)
        cps = _ceil_div(nch, want)
        return _ceil_div(nch, cps)
    tiles = _ceil_div(Cout, 32) * (K // 32)
    # minibatch-sized launches: enough workgroups to fill the chip, and at most 16 chunks walked per
    # workgroup
    want = min(max(448 // tiles, _ceil_div(nch, _WGRAD_CPS), 1), nch, 1024)
    cps = _ceil_div(nch, want)
    return _ceil_div(nch, cps)


# (measurement hooks of tools/layer_bench.py)
_WGRAD_CPS = int(os.environ.get("PFRL_WGRAD_CPS", "16"))
_WGRAD_MAX_SPLITS = int(os.environ.get("PFRL_WGRAD_MAX_SPLITS", "4096"))


_FUSE_BWD = os.environ.get("PFRL_FUSE_BWD", "1") != "0"


def _fused_bwd_ok(N, H, W, C, ST):
    """Both gradients of a layer in one launch: only where the input-gradient kernel runs
    one of its small tile programs (the rule of dgrad_program() in csrc/qnet.hip)."""
    if not _FUSE_BWD:
        return False
    mc = N * (H // ST) * (W // ST)
    z = ST * ST
    if C % 32 == 0 and _ceil_div(mc, 64) * (C // 32) * z >= 1024:
        return False
    if C % 64 == 0 and _ceil_div(mc, 64) * (C // 64) * z >= 1024:
        return False
    return True


# The hidden layer's weight gradient formed INSIDE the optimizer launch (GradSource.lowrank) is
# off by default: measured on MI355X (profiles/r03_fused_optimizer.txt) the weight-gradient half
# of the hidden layer's fused backward launch is nearly free (dgrad + wgrad 14.0 us, dgrad alone
# 12.2 us), while the optimizer launch grows from 13 to 22-34 us when it has to build the tiles.
_LOWRANK = os.environ.get("PFRL_FUSED_OPT_LOWRANK", "0") == "1"


def _lowrank_ok(M, F, K, w):
    from pfrl_amd.optimizers import GradSource

    return w.is_contiguous() and GradSource.lowrank_supported(M, F, K)


def _dist_initialized():
    d = torch.distributed
    return d.is_available() and d.is_initialized()


# A fold of this many slabs or more runs in two levels (pfrl_splitk_group): the weight gradient of
# the first convolution at rollout size arrives as 1 600 - 4 096 slabs of 33 KB, which n / 1024 = 9
# workgroups would otherwise walk eight at a time.  Minibatch-sized folds (<= 50 slabs) keep their
# summation order (the bit-identity tests of the replay agents' update pin it).
_TWO_LEVEL_FOLD_MIN = int(os.environ.get("PFRL_TWO_LEVEL_FOLD_MIN", "256"))


def _pre_fold(tasks):
    """Tasks that share one slab buffer with very many slabs: one grouped launch sums ~sqrt(S)
    consecutive slabs each into a small buffer, and the tasks are re-pointed at that."""
    out = list(tasks)
    done = {}
    for i, t in enumerate(out):
        part, _, _, stride, n, splits = t[:6]
        if splits < _TWO_LEVEL_FOLD_MIN or stride <= 0 or stride % 4:
            continue
        base = part.data_ptr()
        # (weight and bias slabs of one layer live in the same buffer: fold whole slab rows once)
        key = next((k for k in done if k[1] == stride and k[2] == splits
                    and 0 <= base - k[0] < 4 * stride), None)
        if key is None:
            width = stride
            groups = max(8, int(round(splits ** 0.5)))
            per = _ceil_div(splits, groups)
            groups = _ceil_div(splits, per)
            tmp = torch.empty(groups * width, dtype=torch.float32, device=part.device)
            check(_native.lib().pfrl_splitk_group(_p(part), stride, width, splits, groups, _p(tmp), width,
                                                  _stream()), "splitk_group")
            key = (base, stride, splits)
            done[key] = (tmp, groups, width)
        tmp, groups, width = done[key]
        off = (base - key[0]) // 4
        out[i] = (tmp[off:], t[1], t[2], width, n, groups) + tuple(t[6:])
    return out


def _reduce(tasks):
    """tasks: (part, out, bias or None, stride, n, splits, ncol, relu)"""
    if any(t[5] >= _TWO_LEVEL_FOLD_MIN for t in tasks):
        tasks = _pre_fold(tasks)
    n = len(tasks)
    L = _native.lib()
    P = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in tasks])
    O = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in tasks])
    B = (ctypes.c_void_p * n)(*[t[2].data_ptr() if t[2] is not None else 0 for t in tasks])
    S = (ctypes.c_int64 * n)(*[t[3] for t in tasks])
    N = (ctypes.c_int32 * n)(*[t[4] for t in tasks])
    K = (ctypes.c_int32 * n)(*[t[5] for t in tasks])
    C = (ctypes.c_int32 * n)(*[t[6] for t in tasks])
    R = (ctypes.c_int32 * n)(*[t[7] for t in tasks])
    check(L.pfrl_splitk_reduce(n, P, O, B, S, N, K, C, R, _stream()), "splitk_reduce")


# Set to a dict by a caller whose optimizer finishes the gradients itself
# (pfrl_amd.optimizers.FusedRMSprop.step_from_sources, GraphedUpdate): the backward pass then
# hands over split-K slabs and the hidden layer's batch matrices as GradSource objects, keyed by
# the parameter's data_ptr(), instead of folding / forming the gradients -- and returns None as
# the gradient of those parameters.
OPT_SOURCES = None
# An optimizer (FusedRMSprop) whose steps may RIDE in the last backward launch: set next to
# OPT_SOURCES by GraphedUpdate.  The hidden layer's weight and bias (95 % of the parameters; their
# gradients are final after the first backward launch and nothing reads the parameters again in
# the update) are then stepped by extra workgroups of the first convolution's weight-gradient
# launch (pfrl_conv2d_nhwc_bwd_weight_ride) and marked GradSource.done() for the optimizer launch.
RIDE_ALONG = None
_RIDE = os.environ.get("PFRL_RIDE_ALONG", "1") != "0"
# Round 6, measured and NOT the default (PFRL_RIDE_MORE=1 turns it on): EVERY finished layer's step
# riding in the next backward launch (pfrl_ride_set) -- the head's in the last convolution's
# backward launch, each convolution's in the launch of the layer below it, the optimizer's own
# launch left with the first convolution and the loss fold.  Bit-identical
# (tests/test_fused_optimizer.py), and slower: update 84.8 -> 87.8 us, 41.5 -> 40.4 k env-steps/s on
# one box, twice (profiles/r06_ride_more.txt).  The residual launch does not get shorter -- it is
# the 50-slab sum of the first convolution either way, 7 dependent round trips on 8 workgroups --
# while three launches that are latency-bound at 1-3 workgroups per CU each gain slab-summing
# workgroups and 0.5 KB of arguments.
# RIDE_HEAD: {gradient tensor data_ptr: parameter} of the narrow head, set next to RIDE_ALONG by
# GraphedUpdate (the head's per-row partials sit in _DEFERRED_FOLDS).
RIDE_HEAD = None
_RIDE_MORE = os.environ.get("PFRL_RIDE_MORE", "0") == "1"

# dh.data_ptr() -> (dh with the hidden layer's ReLU mask applied and scaled by 1 / world, world):
# left by the fused head + TD-loss launch of a data-parallel update (ops._DQNHeadTDLoss) for the
# low-rank exchange below, which then skips its own mask / scale launches.
MASKED_DH = {}

# Folds queued by other nodes of the same backward pass (the fused head + loss launch of
# ops.dqn_head_td_loss) for the fold launch that ends the trunk's backward: one launch less.
_DEFERRED_FOLDS = []


def defer_fold(tasks):
    _DEFERRED_FOLDS.extend(tasks)


def flush_deferred_folds():
    """Launch whatever is still queued (no MFMA trunk ran a backward after it was queued)."""
    while _DEFERRED_FOLDS:
        batch = _DEFERRED_FOLDS[:12]
        del _DEFERRED_FOLDS[:12]
        _reduce(batch)


def conv_fwd(x, w, b, sp, N, relu=True, planar=False):
    """x: NHWC memory of [N, H, W, C]; returns NHWC [N, OH, OW, Cout] or planar [N, Cout, OH*OW]."""
    shape = (N, sp.Cout, sp.OH * sp.OW) if planar else (N, sp.OH, sp.OW, sp.Cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    check(_native.lib().pfrl_conv2d_nhwc_fwd(_p(x), _p(w), _p(b), _p(y), N, sp.H, sp.W, sp.C, sp.Cout,
                                             sp.R, sp.S, sp.ST, int(relu), int(planar), 1, _stream()),
          "conv2d_nhwc_fwd")
    return y


def conv_fwd_u8(px, w, b, sp, N, relu=True, planar=False):
    """:func:`conv_fwd` for the first layer reading u8 NHWC4 pixels (ops.U8Pixels): phi(x) =
    float32(x) / divisor in the operand loader; bit-identical to conv_fwd on the fp32 minibatch."""
    shape = (N, sp.Cout, sp.OH * sp.OW) if planar else (N, sp.OH, sp.OW, sp.Cout)
    y = torch.empty(shape, dtype=torch.float32, device=px.data.device)
    check(_native.lib().pfrl_conv2d_u8nhwc4_fwd(_p(px.data), px.divisor, _p(w), _p(b), _p(y), N, sp.H,
                                                sp.W, sp.Cout, sp.R, sp.S, sp.ST, int(relu),
                                                int(planar), _stream()), "conv2d_u8nhwc4_fwd")
    return y


def linear_fwd(x, w, b, relu=True):
    """x: [M, K] contiguous -> relu(x w^T + b) [M, F]; split-K + fold for small M."""
    M, K = x.shape
    F = w.shape[0]
    splits = _fwd_splits(M, F, K)
    y = torch.empty((M, F), dtype=torch.float32, device=x.device)
    if splits == 1:
        check(_native.lib().pfrl_conv2d_nhwc_fwd(_p(x), _p(w), _p(b), _p(y), M, 1, 1, K, F, 1, 1, 1,
                                                 int(relu), 0, 1, _stream()), "linear_fwd")
        return y
    part = torch.empty((splits, M, F), dtype=torch.float32, device=x.device)
    check(_native.lib().pfrl_conv2d_nhwc_fwd(_p(x), _p(w), None, _p(part), M, 1, 1, K, F, 1, 1, 1, 0,
                                             0, splits, _stream()), "linear_fwd_splitk")
    if FWD_FOLD_SINK is not None and relu and b is not None:
        # the consumer (the fused head + TD-loss launch) folds the slabs itself and fills y
        FWD_FOLD_SINK[y.data_ptr()] = (y, part, b, M * F, splits)
        return y
    _reduce([(part, y, b, M * F, M * F, splits, F, int(relu))])
    return y


# Set to a dict by a caller that can fold the hidden layer's split-K slabs in its own launch
# (DQN._compute_loss_fused -> ops.dqn_head_td_loss(h_fold=...)): linear_fwd then returns the
# output tensor UNFILLED and records data_ptr -> (y, part, bias, stride, splits) here.  Whatever
# the caller does not consume it must hand to flush_fwd_folds().
FWD_FOLD_SINK = None


def flush_fwd_folds(sink):
    for y, part, b, stride, splits in sink.values():
        _reduce([(part, y, b, stride, stride, splits, y.shape[1], 1)])
    sink.clear()


# Rollout- and update-sized batches: the last convolution writes plain NHWC rows and the linear layer
# reads them through a copy of its weight with the columns re-ordered (c, p) -> (p, c), instead of
# the planar (NCHW) output / planar -> NHWC input gradient of the minibatch route.  Planar stores
# and the permuted input-gradient stores are 4-byte accesses 196 B / 256 B apart: 51 M of them per
# launch at B = 16384 (PPO's minibatch), where they -- not the MFMAs -- set the launch time.  The
# re-ordered weight is a 6.4 MB copy per forward pass, the weight gradient is re-ordered back by
# another.  Not for the replay agents' minibatches, whose bit-identity tests pin the planar route.
_NHWC_FC_MIN_BATCH = int(os.environ.get("PFRL_TRUNK_NHWC_FC_MIN_BATCH", "1024"))


def _reordered_weight(wf, C, P):
    """wf [F, C * P] with columns (c, p) -> [F, P * C] with columns (p, c): one 6.4 MB copy (~10 us)
    per forward pass, against >= 200 us of trunk at the batch sizes that take this route.  Not
    cached: the optimizers and the target-network sync of this package write parameters through raw
    pointers, which a tensor's version counter never sees."""
    F = wf.shape[0]
    return wf.detach().view(F, C, P).transpose(1, 2).contiguous().view(F, P * C)


class _Trunk(torch.autograd.Function):
    """h = relu(linear(flatten(relu(conv_L(... relu(conv_1(x))))))) as one autograd node."""

    @staticmethod
    def forward(ctx, x, specs, *params):
        from pfrl_amd.ops import U8Pixels

        N = x.shape[0]
        u8 = isinstance(x, U8Pixels)      # (the first layer evaluates phi itself: plan_for agreed)
        if not u8 and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        L = len(specs)
        acts = []
        h = x
        nhwc_fc = len(params) > 2 * L and 0 < _NHWC_FC_MIN_BATCH <= max(N, _PLAN_BATCH)
        for i, sp in enumerate(specs):
            fwd = conv_fwd_u8 if (u8 and i == 0) else conv_fwd
            h = fwd(h, params[2 * i], params[2 * i + 1], sp, N, relu=True,
                    planar=(i == L - 1 and not nhwc_fc))
            acts.append(h)
        wp = None
        if len(params) == 2 * L:
            # convolutions only: the planar (NCHW) output of the last one, as [N, Cout, OH, OW]
            out = h.view(N, specs[-1].Cout, specs[-1].OH, specs[-1].OW)
        else:
            wf, bf = params[2 * L], params[2 * L + 1]
            if nhwc_fc:
                wp = _reordered_weight(wf, specs[-1].Cout, specs[-1].OH * specs[-1].OW)
                out = linear_fwd(h.view(N, -1), wp, bf, relu=True)
            else:
                out = linear_fwd(h.view(N, -1), wf, bf, relu=True)
        if any(ctx.needs_input_grad[2:]):
            ctx.specs = specs
            ctx.nhwc_fc = nhwc_fc
            ctx.u8_divisor = x.divisor if u8 else None
            ctx.save_for_backward(x.data if u8 else x, out, *acts, *params,
                                  *([wp] if nhwc_fc else []))
        return out

    @staticmethod
    def backward(ctx, dh):
        specs = ctx.specs
        L = len(specs)
        saved = ctx.saved_tensors
        x, out = saved[0], saved[1]
        acts = saved[2:2 + L]
        params = saved[2 + L:]
        nhwc_fc = getattr(ctx, "nhwc_fc", False)
        if nhwc_fc:
            params, wp = params[:-1], params[-1]
        N = x.shape[0]
        lib = _native.lib()
        dev = x.device
        dh = dh.contiguous()
        last = specs[-1]
        P = last.OH * last.OW
        if len(params) == 2 * L:
            # convolutions only: ReLU mask of the last convolution and planar -> NHWC rows
            g = torch.ops.aten.threshold_backward(dh.view(N, last.Cout, P), acts[-1].view(N, last.Cout, P),
                                                  0.0)
            dy = g.permute(0, 2, 1).contiguous().view(N, last.OH, last.OW, last.Cout)
            return _Trunk._conv_backward(ctx, specs, params, acts, x, dy, N, dev, [None] * (2 * L))
        wf = params[2 * L]
        F, Kf = wf.shape
        # hidden linear layer: input gradient straight into NHWC rows of the last conv, and
        # the weight gradient, in one launch when the batch is minibatch-sized
        dy = torch.empty((N, last.OH, last.OW, last.Cout), dtype=torch.float32, device=dev)
        if nhwc_fc:
            # everything in (p, c) column order: plain NHWC rows in, NHWC rows out, the mask of the
            # convolution below read where the gradient is written; the weight gradient goes back
            # to the parameter's (c, p) order with one copy
            # (the ReLU mask of the hidden layer applied once, by one elementwise launch, instead of
            # being loaded beside dh by both gradient kernels: at this size the second operand
            # stream costs each of them more than the extra launch)
            dhm = torch.ops.aten.threshold_backward(dh, out, 0.0)
            check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dhm), None, _p(wp), _p(acts[-1]), _p(dy), N, 1, 1,
                                                Kf, F, 1, 1, 1, 0, 0, _stream()), "linear_bwd_data")
            dwp = torch.empty_like(wp)
            dbf = torch.empty(F, dtype=torch.float32, device=dev)
            fsplits = _wgrad_splits(N, F, Kf)
            if fsplits == 1:
                check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dhm), None, _p(acts[-1]), _p(dwp), _p(dbf), 0,
                                                      0, N, 1, 1, Kf, F, 1, 1, 1, 1, _stream()),
                      "linear_bwd_weight")
            else:
                fstride = F * Kf + F
                fpart = torch.empty(fsplits * fstride, dtype=torch.float32, device=dev)
                check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dhm), None, _p(acts[-1]), _p(fpart),
                                                      _p(fpart[F * Kf:]), fstride, fstride, N, 1, 1, Kf,
                                                      F, 1, 1, 1, fsplits, _stream()),
                      "linear_bwd_weight")
                _reduce([(fpart, dwp, None, fstride, F * Kf, fsplits, 4, 0),
                         (fpart[F * Kf:], dbf, None, fstride, F, fsplits, 4, 0)])
            dwf = dwp.view(F, P, last.Cout).transpose(1, 2).contiguous().view(F, Kf)
            if _dist_initialized():
                from pfrl_amd.distributed import announce_grad

                announce_grad(wf, dwf)
            grads = [None] * (2 * L) + [dwf, dbf]
            return _Trunk._conv_backward(ctx, specs, params, acts, x, dy, N, dev, grads)
        if OPT_SOURCES is not None and _LOWRANK and _lowrank_ok(N, F, Kf, wf):
            # the optimizer forms dW = dh^T x itself, tile by tile, and applies it from the
            # accumulators (csrc/optim.hip): only the input gradient is computed here
            from pfrl_amd.optimizers import GradSource

            check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dh), _p(out), _p(wf), _p(acts[-1]), _p(dy), N, 1, 1,
                                                Kf, F, 1, 1, 1, P, last.Cout, _stream()),
                  "linear_bwd_data")
            OPT_SOURCES[wf.data_ptr()] = GradSource.lowrank(dh, out, acts[-1].view(N, Kf))
            OPT_SOURCES[params[2 * L + 1].data_ptr()] = GradSource.lowrank_bias(dh, out)
            grads = [None] * (2 * L + 2)
            return _Trunk._conv_backward(ctx, specs, params, acts, x, dy, N, dev, grads)
        if _dist_initialized():
            from pfrl_amd.distributed import announce_grad, announce_lowrank, lowrank_wanted

            if lowrank_wanted(wf, N):
                # data parallel at minibatch size: the layer's weight gradient is not formed here
                # at all.  Its batch matrices (dh with the ReLU mask applied, and the layer's input)
                # leave NOW for an all-gather under the convolution backward below; every rank
                # then forms dW = sum_g dh_g^T x_g / G itself (distributed.py: 0.46 MB per rank on
                # the links instead of a 6.4 MB all-reduce)
                # (mask, scale, all-gathers, product: all on the communicator's side stream; this
                # stream goes straight on with the input gradient, mask applied in the kernel)
                # The input gradient goes first: it is the last reader of this layer's weight in the
                # update, and the side stream -- forked behind it -- may step the weight.
                check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dh), _p(out), _p(wf), _p(acts[-1]), _p(dy), N, 1,
                                                    1, Kf, F, 1, 1, 1, P, last.Cout, _stream()),
                      "linear_bwd_data")
                grads = [None] * (2 * L + 2)
                if not announce_lowrank(wf, params[2 * L + 1], dh, acts[-1].view(N, Kf), mask=out,
                                        premasked=MASKED_DH.pop(dh.data_ptr(), None)):
                    # (not taken after all, e.g. a capture without collectives: the local gradient)
                    dwf = torch.empty_like(wf)
                    dbf = torch.empty(F, dtype=torch.float32, device=dev)
                    check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dh), _p(out), _p(acts[-1].view(N, Kf)),
                                                          _p(dwf), _p(dbf), 0, 0, N, 1, 1, Kf, F, 1, 1, 1,
                                                          1, _stream()), "linear_bwd_weight")
                    announce_grad(wf, dwf)
                    grads[2 * L], grads[2 * L + 1] = dwf, dbf
                return _Trunk._conv_backward(ctx, specs, params, acts, x, dy, N, dev, grads)
        dwf = torch.empty_like(wf)
        dbf = torch.empty(F, dtype=torch.float32, device=dev)
        if _fused_bwd_ok(N, 1, 1, Kf, 1):
            check(lib.pfrl_conv2d_nhwc_bwd(_p(dh), _p(out), _p(wf), _p(acts[-1]), _p(acts[-1]), _p(dy),
                                           _p(dwf), _p(dbf), 0, 0, N, 1, 1, Kf, F, 1, 1, 1, P,
                                           last.Cout, 1, _stream()), "linear_bwd")
        else:
            check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dh), _p(out), _p(wf), _p(acts[-1]), _p(dy), N, 1, 1,
                                                Kf, F, 1, 1, 1, P, last.Cout, _stream()),
                  "linear_bwd_data")
            check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dh), _p(out), _p(acts[-1]), _p(dwf), _p(dbf), 0, 0,
                                                  N, 1, 1, Kf, F, 1, 1, 1, 1, _stream()),
                  "linear_bwd_weight")
        if _dist_initialized():
            # data parallel: the hidden layer's gradient (95 % of the bytes) starts its
            # all-reduce now, under the convolution backward that follows
            from pfrl_amd.distributed import announce_grad

            announce_grad(wf, dwf)
        grads = [None] * (2 * L) + [dwf, dbf]
        ride = None
        if (RIDE_ALONG is not None and OPT_SOURCES is not None and _RIDE and L >= 1
                and not _dist_initialized() and getattr(ctx, "u8_divisor", None) is None):
            ride = [(wf, dwf, 2 * L), (params[2 * L + 1], dbf, 2 * L + 1)]
        return _Trunk._conv_backward(ctx, specs, params, acts, x, dy, N, dev, grads, ride)

    @staticmethod
    def _conv_backward(ctx, specs, params, acts, x, dy, N, dev, grads, ride=None):
        """Gradients of the convolutions given dy = dL/d(output of the last one) as NHWC rows
        (its ReLU mask applied)."""
        lib = _native.lib()
        L = len(specs)
        tasks = []
        # steps waiting for the next backward launch: [(parameter, dense gradient or GradSource)]
        riding = (ride is not None and _RIDE_MORE and RIDE_ALONG is not None and OPT_SOURCES is not None
                  and hasattr(RIDE_ALONG, "ride_set"))
        carry = []
        if riding and RIDE_HEAD and _DEFERRED_FOLDS:
            from pfrl_amd.optimizers import GradSource

            for t in list(_DEFERRED_FOLDS):
                p_ = RIDE_HEAD.get(t[1].data_ptr())
                if p_ is not None and len(carry) < 4:
                    carry.append((p_, GradSource.slabs(t[0], t[3], t[5]), t))
        for i in range(L - 1, -1, -1):
            sp = specs[i]
            w = params[2 * i]
            below = acts[i - 1] if i > 0 else x
            nW = w.numel()
            M = N * sp.OH * sp.OW
            splits = _wgrad_splits(M, sp.Cout, sp.R * sp.S * sp.C)
            stride = nW + sp.Cout
            dw = torch.empty_like(w)
            db = torch.empty(sp.Cout, dtype=torch.float32, device=dev)
            if splits == 1:
                pw, pb, st = dw, db, 0
            else:
                part = torch.empty(splits * stride, dtype=torch.float32, device=dev)
                pw, pb, st = part, part[nW:], stride
                tasks.append((part, dw, None, stride, nW, splits, 4, 0))
                tasks.append((pb, db, None, stride, sp.Cout, splits, 4, 0))
            grads[2 * i], grads[2 * i + 1] = dw, db
            if OPT_SOURCES is not None and splits > 1:
                from pfrl_amd.optimizers import GradSource

                # the slabs go to the optimizer as they are (no fold, no gradient tensor)
                del tasks[-2:]
                OPT_SOURCES[w.data_ptr()] = GradSource.slabs(part, stride, splits)
                OPT_SOURCES[params[2 * i + 1].data_ptr()] = GradSource.slabs(pb, stride, splits)
                grads[2 * i], grads[2 * i + 1] = None, None
            if i > 0 and _fused_bwd_ok(N, sp.H, sp.W, sp.C, sp.ST):
                dx = torch.empty((N, sp.H, sp.W, sp.C), dtype=torch.float32, device=dev)
                bwd_args = (_p(dy), None, _p(w), _p(below), _p(below), _p(dx), _p(pw), _p(pb), st, st, N,
                            sp.H, sp.W, sp.C, sp.Cout, sp.R, sp.S, sp.ST, 0, 0, splits, _stream())
                rode = False
                if carry and RIDE_ALONG.ride_set([(p_, g_) for p_, g_, _ in carry]):
                    # (a tile program without a riding form refuses BEFORE launching and drops the set)
                    rode = lib.pfrl_conv2d_nhwc_bwd(*bwd_args) == 0
                if not rode:
                    check(lib.pfrl_conv2d_nhwc_bwd(*bwd_args), "conv2d_nhwc_bwd")
                carry = _after_ride(carry, rode)
                if riding and splits > 1 and OPT_SOURCES is not None:
                    # this layer's own step: in the launch of the layer below
                    carry += [(w, OPT_SOURCES[w.data_ptr()], None),
                              (params[2 * i + 1], OPT_SOURCES[params[2 * i + 1].data_ptr()], None)]
                dy = dx
                continue
            ra = RIDE_ALONG.ride_arrays([(p_, g_) for p_, g_, _ in ride]) if (ride and i == 0) else None
            if ra is not None:
                # the last launch of the backward pass: the finished layers' optimizer steps ride in it
                from pfrl_amd.optimizers import GradSource

                more = bool(carry) and len(carry) + len(ride) <= 8 and \
                    RIDE_ALONG.ride_set([(p_, g_) for p_, g_, _ in carry])
                check(lib.pfrl_conv2d_nhwc_bwd_weight_ride(
                    _p(dy), None, _p(below), _p(pw), _p(pb), st, st, N, sp.H, sp.W, sp.C, sp.Cout, sp.R,
                    sp.S, sp.ST, splits, *ra, _stream()), "conv2d_nhwc_bwd_weight_ride")
                carry = _after_ride(carry, more)
                for p_, g_, slot in ride:
                    OPT_SOURCES[p_.data_ptr()] = GradSource.done()
                    grads[slot] = None
                continue
            u8d = getattr(ctx, "u8_divisor", None)
            if i == 0 and u8d is not None:
                # the layer input is the u8 minibatch itself (saved as such: a quarter of the bytes)
                check(lib.pfrl_conv2d_u8nhwc4_bwd_weight(_p(dy), None, _p(below), u8d, _p(pw), _p(pb),
                                                         st, st, N, sp.H, sp.W, sp.Cout, sp.R, sp.S,
                                                         sp.ST, splits, _stream()),
                      "conv2d_u8nhwc4_bwd_weight")
                continue
            check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dy), None, _p(below), _p(pw), _p(pb), st, st, N,
                                                  sp.H, sp.W, sp.C, sp.Cout, sp.R, sp.S, sp.ST, splits,
                                                  _stream()), "conv2d_nhwc_bwd_weight")
            if i > 0:
                dx = torch.empty((N, sp.H, sp.W, sp.C), dtype=torch.float32, device=dev)
                check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dy), None, _p(w), _p(below), _p(dx), N, sp.H,
                                                    sp.W, sp.C, sp.Cout, sp.R, sp.S, sp.ST, 0, 0,
                                                    _stream()), "conv2d_nhwc_bwd_data")
                dy = dx
        if OPT_SOURCES is None and _DEFERRED_FOLDS and len(tasks) + len(_DEFERRED_FOLDS) <= 12:
            tasks += _DEFERRED_FOLDS
            del _DEFERRED_FOLDS[:]
        if tasks:
            _reduce(tasks)
        return (None, None) + tuple(grads)


def _after_ride(carry, rode):
    """Bookkeeping behind a backward launch that carried (``rode``) or did not carry the steps of
    ``carry``: carried parameters are marked done for the optimizer launch, and the head's
    deferred folds they came from are withdrawn.  Returns the new (empty) carry list; what did not
    ride stays with the optimizer launch, as before."""
    if rode:
        from pfrl_amd.optimizers import GradSource

        for p_, _, task in carry:
            OPT_SOURCES[p_.data_ptr()] = GradSource.done()
            if task is not None and task in _DEFERRED_FOLDS:
                _DEFERRED_FOLDS.remove(task)
    return []


def trunk_forward(x, specs, convs, linear):
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    if linear is not None:
        params += [linear.weight, linear.bias]
    return _Trunk.apply(x, specs, *params)


class _SmallLinear(torch.autograd.Function):
    """y = x w^T + b for a narrow head (out_features <= 16): one launch forward, one
    launch for (dx, dw, db)."""

    @staticmethod
    def forward(ctx, x, w, b):
        M, K = x.shape
        Nout = w.shape[0]
        x = x.contiguous()
        y = torch.empty((M, Nout), dtype=torch.float32, device=x.device)
        check(_native.lib().pfrl_linear_small_fwd(_p(x), _p(w), _p(b), _p(y), M, K, Nout, _stream()),
              "linear_small_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, K = x.shape
        Nout = w.shape[0]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if not ctx.needs_input_grad[1]:
            # frozen weights: the input gradient only
            if dx is None:
                return None, None, None
            check(_native.lib().pfrl_linear_small_bwd(_p(dy), _p(x), _p(w), _p(dx), None, None, M, K,
                                                      Nout, _stream()), "linear_small_bwd")
            return dx, None, None
        dw = torch.empty_like(w)
        db = torch.empty(Nout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        check(_native.lib().pfrl_linear_small_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), M, K,
                                                  Nout, _stream()), "linear_small_bwd")
        return dx, dw, db


def small_linear_supported(layer, x):
    return (_ENABLED and isinstance(layer, nn.Linear) and x.is_cuda and x.dim() == 2
            and x.dtype == torch.float32 and layer.out_features <= SMALL_LINEAR_MAX_OUT
            and layer.weight.dtype == torch.float32 and layer.weight.is_contiguous()
            and x.shape[0] * layer.out_features <= 10240 and x.shape[0] <= 4096
            and _native.available())


def small_linear(x, layer):
    return _SmallLinear.apply(x, layer.weight, layer.bias)


class _SmallLinearSlot(nn.Linear):
    """An ``nn.Linear`` (same class, same parameter names, same state_dict) whose forward
    takes the narrow-head kernels on the GPU; built around the tensors of an existing layer."""

    def __init__(self, linear):
        nn.Module.__init__(self)
        self.__dict__.update({k: v for k, v in linear.__dict__.items()
                              if k not in ("_parameters", "_buffers", "_modules")})
        self._parameters = linear._parameters
        self._buffers = linear._buffers
        self._modules = linear._modules

    def forward(self, x):
        if small_linear_supported(self, x):
            return small_linear(x, self)
        return nn.Linear.forward(self, x)


class _TrunkSequential(nn.Sequential):
    """An ``nn.Sequential`` (same children, same indices, same state_dict) whose forward runs a
    ``(Conv2d, ReLU)+, Flatten, Linear, ReLU`` stretch of its children as the MFMA trunk when
    the input is inside what the kernels cover, and child by child otherwise."""

    def forward(self, x):
        from pfrl_amd.ops import U8Pixels

        start, end, conv_idx, lin_idx = self._trunk_run
        mods = list(self._modules.values())
        if isinstance(x, U8Pixels) and (start != 0 or plan_for([mods[k] for k in conv_idx],
                                                                mods[lin_idx], x) is None):
            x = x.float()       # nobody here reads u8 pixels: the fp32 input they stand for
        i = 0
        while i < len(mods):
            if i == start:
                convs = [mods[k] for k in conv_idx]
                specs = plan_for(convs, mods[lin_idx], x)
                if specs is not None:
                    x = trunk_forward(x, specs, convs, mods[lin_idx])
                    i = end
                    if FWD_FOLD_SINK and i < len(mods):
                        # slabs may only be handed to the narrow head behind this container; a
                        # child that follows the stretch reads the folded tensor
                        flush_fwd_folds(FWD_FOLD_SINK)
                    continue
            x = mods[i](x)
            i += 1
        return x


def _find_trunk_run(mods):
    """(start, end, conv indices, linear index) of the first ``(Conv2d, ReLU)+, Flatten, Linear,
    ReLU`` stretch in a list of modules (a ReLU slot may hold the identity that
    ``fuse_conv_bias_relu`` leaves behind), or None."""
    from pfrl_amd.nn.atari_cnn import _Identity

    relu_like = (nn.ReLU, _Identity, nn.Identity)
    for start in range(len(mods)):
        j, convs = start, []
        while (j + 1 < len(mods) and isinstance(mods[j], nn.Conv2d)
               and isinstance(mods[j + 1], relu_like)):
            convs.append(j)
            j += 2
        if (convs and j + 2 < len(mods) + 0 and isinstance(mods[j], nn.Flatten)
                and isinstance(mods[j + 1], nn.Linear) and isinstance(mods[j + 2], nn.ReLU)):
            return start, j + 3, convs, j + 1
    return None


def fuse_sequential_trunk(model):
    """Make every ``nn.Sequential`` inside ``model`` that contains a ``(Conv2d, ReLU)+, Flatten,
    Linear, ReLU`` stretch (the PPO / A2C example networks, examples/atari/train_ppo_ale.py:247-264)
    execute that stretch as the MFMA trunk.  Children, parameters and state_dict keys are
    untouched: only the container's class changes."""
    for seq in [m for m in model.modules() if type(m) is nn.Sequential]:
        run = _find_trunk_run(list(seq._modules.values()))
        if run is not None:
            seq.__class__ = _TrunkSequential
            object.__setattr__(seq, "_trunk_run", run)
    return model


def accelerate_heads(model):
    """Replace every narrow ``nn.Linear`` (out_features <= 16, plain class) inside ``model``
    by a ``_SmallLinearSlot`` sharing its parameters."""
    for parent in list(model.modules()):
        for name, child in list(parent._modules.items()):
            if type(child) is nn.Linear and child.out_features <= SMALL_LINEAR_MAX_OUT:
                parent._modules[name] = _SmallLinearSlot(child)
    return model
