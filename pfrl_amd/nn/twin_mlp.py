"""The twin Q-networks of SAC / TD3 evaluated as ONE chain of launches.

``q_func1`` and ``q_func2`` (``pfrl/agents/soft_actor_critic.py:97-110``, the 256-256 MLPs of
``examples/mujoco/reproduction/soft_actor_critic/train_soft_actor_critic.py:172-199``) are two
independent networks of the same shape that the update always evaluates on the same input,
one after the other: six forward passes and four backward chains per update, each kernel a
~4 us launch inside the captured graph for well under 1 us of work.  ``twin_forward`` runs
each layer of both networks as one grid (``pfrl_linear_fwd_twin`` and friends in
``csrc/qnet.hip``: blockIdx.z / .y picks the network) as a single autograd node:

    forward   cat, layer 1 (any in_features), layer 2, narrow head              4 launches for both
    backward  head, layer 2 (dx + dw), layer 1 (dw), one fold of all partials   4 launches for both
    frozen    (requires_grad off on the parameters: the policy loss) head dx, layer 2 dx and
              the gradient w.r.t. the action columns only                     3 launches for both

The arithmetic per network is that of the single-network kernels (``mfma_linear``); only the
action-column input gradient is a different (smaller) sum than ``dy @ W`` followed by a slice.

Applies when both modules are ``accelerate_mlp``-ed ``nn.Sequential(ConcatObsAndAction(),
Linear, ReLU, Linear, ReLU, Linear(out <= 16))`` with equal shapes, hidden widths multiples of
32, float32 on the GPU; ``twin_forward`` returns None otherwise and the caller evaluates the two
modules one by one.
"""
import ctypes

import torch
import torch.nn as nn

from pfrl_amd import _native
from pfrl_amd._native import check
from pfrl_amd.nn import mfma_linear as _ml
from pfrl_amd.nn import mfma_trunk as _t
from pfrl_amd.nn.concat_obs_and_action import ConcatObsAndAction
from pfrl_amd.nn.mfma_trunk import _ceil_div, _stream

MAX_BATCH = 512


def _pair(a, b):
    return (ctypes.c_void_p * 2)(a.data_ptr() if a is not None else 0,
                                 b.data_ptr() if b is not None else 0)


def _layers(q):
    mods = list(q._modules.values()) if isinstance(q, nn.Sequential) else None
    if mods is None or len(mods) != 6:
        return None
    c, l1, r1, l2, r2, l3 = mods
    if not (type(c) is ConcatObsAndAction and isinstance(l1, _ml._LinearSlot) and type(r1) is nn.ReLU
            and isinstance(l2, _ml._LinearSlot) and type(r2) is nn.ReLU
            and isinstance(l3, _t._SmallLinearSlot)):
        return None
    return l1, l2, l3


def twin_plan(q1, q2, inputs):
    """The six layers if the pair can run twinned on ``inputs = (obs, action)``, else None."""
    if not (_ml._ENABLED and _native.available() and isinstance(inputs, (tuple, list)) and len(inputs) == 2):
        return None
    obs, action = inputs
    if not (torch.is_tensor(obs) and torch.is_tensor(action) and obs.is_cuda and action.is_cuda
            and obs.dtype == torch.float32 and action.dtype == torch.float32 and obs.dim() == 2
            and action.dim() == 2 and obs.shape[0] == action.shape[0]
            and 0 < obs.shape[0] <= MAX_BATCH and not obs.requires_grad):
        return None
    la, lb = _layers(q1), _layers(q2)
    if la is None or lb is None:
        return None
    for a, b in zip(la, lb):
        if (a.weight.shape != b.weight.shape or a.bias is None or b.bias is None
                or a.weight.dtype != torch.float32 or b.weight.dtype != torch.float32
                or not a.weight.is_cuda or not b.weight.is_cuda
                or not a.weight.is_contiguous() or not b.weight.is_contiguous()):
            return None
    l1, l2, l3 = la
    if (l1.in_features != obs.shape[1] + action.shape[1] or l1.out_features % 32 or l2.out_features % 32
            or l2.in_features != l1.out_features or l3.in_features != l2.out_features
            or l3.out_features > _t.SMALL_LINEAR_MAX_OUT
            or obs.shape[0] * l3.out_features > 10240):
        return None
    # requires_grad must agree across the pair (all trainable, or all frozen)
    flags = {p.requires_grad for layer in la + lb for p in (layer.weight, layer.bias)}
    if len(flags) != 1:
        return None
    return la + lb


class _TwinQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, action, *params):
        w1a, b1a, w2a, b2a, w3a, b3a, w1b, b1b, w2b, b2b, w3b, b3b = params
        lib = _native.lib()
        dev = obs.device
        M = obs.shape[0]
        Ko, K = obs.shape[1], obs.shape[1] + action.shape[1]
        H1, H2, No = w1a.shape[0], w2a.shape[0], w3a.shape[0]
        h1 = torch.empty((2, M, H1), dtype=torch.float32, device=dev)
        h2 = torch.empty((2, M, H2), dtype=torch.float32, device=dev)
        q = torch.empty((2, M, No), dtype=torch.float32, device=dev)
        st = _stream()
        # layer 1 reads its input rows from obs and action directly (no cat)
        check(lib.pfrl_linear_fwd_twin(_pair(obs, obs), _pair(action, action), Ko, _pair(w1a, w1b),
                                       _pair(b1a, b1b), _pair(h1[0], h1[1]), M, K, H1, 1, st),
              "linear_fwd_twin")
        check(lib.pfrl_linear_fwd_twin(_pair(h1[0], h1[1]), None, 0, _pair(w2a, w2b), _pair(b2a, b2b),
                                       _pair(h2[0], h2[1]), M, H1, H2, 1, st), "linear_fwd_twin")
        check(lib.pfrl_linear_small_fwd_twin(_pair(h2[0], h2[1]), _pair(w3a, w3b), _pair(b3a, b3b),
                                             _pair(q[0], q[1]), M, H2, No, st), "linear_small_fwd_twin")
        ctx.save_for_backward(obs, action, h1, h2, w1a, w2a, w3a, w1b, w2b, w3b)
        ctx.ptrs = [[t.data_ptr() for t in (w1a, b1a, w2a, b2a)], [t.data_ptr() for t in (w1b, b1b, w2b, b2b)]]
        return q[0], q[1]

    @staticmethod
    def backward(ctx, ga, gb):
        obs, action, h1, h2, w1a, w2a, w3a, w1b, w2b, w3b = ctx.saved_tensors
        lib = _native.lib()
        dev = obs.device
        M = obs.shape[0]
        Ko, K = obs.shape[1], obs.shape[1] + action.shape[1]
        H1, H2, No = w1a.shape[0], w2a.shape[0], w3a.shape[0]
        need_da = ctx.needs_input_grad[1]
        need_w = any(ctx.needs_input_grad[2:])
        zeros = None
        if ga is None or gb is None:
            zeros = torch.zeros((M, No), dtype=torch.float32, device=dev)
        ga = ga.contiguous() if ga is not None else zeros
        gb = gb.contiguous() if gb is not None else zeros
        st = _stream()
        f32 = dict(dtype=torch.float32, device=dev)
        dh2 = torch.empty((2, M, H2), **f32)
        dh1 = torch.empty((2, M, H1), **f32)
        grads = [None] * 12
        if need_w:
            dw3 = torch.empty((2,) + tuple(w3a.shape), **f32)
            db3 = torch.empty((2, No), **f32)
            check(lib.pfrl_linear_small_bwd_twin(_pair(ga, gb), _pair(h2[0], h2[1]), _pair(w3a, w3b),
                                                 _pair(dh2[0], dh2[1]), _pair(dw3[0], dw3[1]),
                                                 _pair(db3[0], db3[1]), M, H2, No, st),
                  "linear_small_bwd_twin")
            # layer 2: all four gradients in one launch; layer 1: weight gradients
            s2 = _t._wgrad_splits(M, H2, H1)
            s1 = _t._wgrad_splits(M, H1, _ceil_div(K, 32) * 32)
            n2, n1 = H2 * H1, H1 * K
            st2, st1 = n2 + H2, n1 + H1
            p2 = torch.empty((2, s2 * st2), **f32)
            p1 = torch.empty((2, s1 * st1), **f32)
            check(lib.pfrl_linear_bwd_twin(_pair(dh2[0], dh2[1]), _pair(h2[0], h2[1]), _pair(w2a, w2b),
                                           _pair(h1[0], h1[1]), None, 0, _pair(dh1[0], dh1[1]),
                                           _pair(p2[0], p2[1]), _pair(p2[0][n2:], p2[1][n2:]), st2, st2, M,
                                           H1, H2, s2, st),
                  "linear_bwd_twin")
            check(lib.pfrl_linear_bwd_twin(_pair(dh1[0], dh1[1]), _pair(h1[0], h1[1]), _pair(w1a, w1b),
                                           _pair(obs, obs), _pair(action, action), Ko, None,
                                           _pair(p1[0], p1[1]), _pair(p1[0][n1:], p1[1][n1:]), st1, st1, M,
                                           K, H1, s1, st),
                  "linear_bwd_twin")
            from pfrl_amd.nn import mfma_linear as _ml

            sink = _ml._SLAB_SINK
            if sink is not None and s1 > 1 and s2 > 1:
                # the optimizer sums the slabs inside its own launch (mfma_linear.slab_sink)
                for t in range(2):
                    pw1, pb1, pw2, pb2 = ctx.ptrs[t]
                    sink[pw1], sink[pb1] = (p1[t], st1, s1), (p1[t][n1:], st1, s1)
                    sink[pw2], sink[pb2] = (p2[t], st2, s2), (p2[t][n2:], st2, s2)
                    grads[6 * t:6 * t + 6] = [None, None, None, None, dw3[t], db3[t]]
            else:
                dw2 = torch.empty((2, H2, H1), **f32)
                db2 = torch.empty((2, H2), **f32)
                dw1 = torch.empty((2, H1, K), **f32)
                db1 = torch.empty((2, H1), **f32)
                tasks = []
                for t in range(2):
                    tasks += [(p2[t], dw2[t], None, st2, n2, s2, 4, 0), (p2[t][n2:], db2[t], None, st2, H2, s2, 4, 0),
                              (p1[t], dw1[t], None, st1, n1, s1, 4, 0), (p1[t][n1:], db1[t], None, st1, H1, s1, 4, 0)]
                _t._reduce(tasks)
                for t in range(2):
                    grads[6 * t:6 * t + 6] = [dw1[t], db1[t], dw2[t], db2[t], dw3[t], db3[t]]
        else:
            check(lib.pfrl_linear_small_bwd_twin(_pair(ga, gb), _pair(h2[0], h2[1]), _pair(w3a, w3b),
                                                 _pair(dh2[0], dh2[1]), None, None, M, H2, No, st),
                  "linear_small_bwd_twin")
            if need_da:
                check(lib.pfrl_linear_bwd_twin(_pair(dh2[0], dh2[1]), _pair(h2[0], h2[1]), _pair(w2a, w2b),
                                               None, None, 0, _pair(dh1[0], dh1[1]), None, None, 0, 0, M, H1,
                                               H2, 1, st), "linear_bwd_twin")
        da = None
        if need_da:
            A = K - Ko
            if A <= 32 and H1 <= 512 and 2 * H1 * A * 4 <= 60 * 1024:
                da = torch.empty((M, A), **f32)
                check(lib.pfrl_twin_input_grad(_pair(dh1[0], dh1[1]), _pair(h1[0], h1[1]), _pair(w1a, w1b),
                                               K, Ko, A, ctx_ptr(da), M, H1, st), "twin_input_grad")
            else:
                g = torch.ops.aten.threshold_backward(dh1, h1, 0.0)
                da = g[0] @ w1a[:, Ko:] + g[1] @ w1b[:, Ko:]
        return (None, da) + tuple(grads)


def ctx_ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def twin_forward(q1, q2, inputs):
    """``(q1(inputs), q2(inputs))`` through the twin kernels, or None when the pair is outside
    what they cover."""
    layers = twin_plan(q1, q2, inputs)
    if layers is None:
        return None
    obs, action = inputs
    params = []
    for layer in layers:
        params += [layer.weight, layer.bias]
    return _TwinQ.apply(obs.contiguous(), action.contiguous(), *params)
