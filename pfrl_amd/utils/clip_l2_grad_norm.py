"""Gradient clipping by global L2 norm (reference
pfrl/utils/clip_l2_grad_norm.py:5-38)."""
import ctypes

import torch


def clip_l2_grad_norm_(parameters, max_norm):
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    return torch.nn.utils.clip_grad_norm_(list(parameters), float(max_norm), norm_type=2)


def clip_grad_norm_device_(parameters, max_norm):
    """``torch.nn.utils.clip_grad_norm_(parameters, max_norm)`` for dense f32 CUDA gradients in
    three launches (pfrl_clip_grad_norm) instead of ~12; anything else takes the torch function.
    Returns the total norm (a 0-dim tensor), as torch does."""
    params = [p for p in parameters if p.grad is not None]
    grads = [p.grad for p in params]

    def dense(g):
        # (any permutation of a contiguous layout, e.g. channels_last convolution weights: the
        # norm and the scaling are elementwise, memory order is irrelevant)
        return g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last) \
            or g.permute(*sorted(range(g.dim()), key=lambda d: -g.stride(d))).is_contiguous()

    if not (0 < len(grads) <= 24 and max_norm is not None and float(max_norm) > 0
            and all(g.is_cuda and g.dtype == torch.float32 and g.numel() > 0 and dense(g)
                    for g in grads)):
        return torch.nn.utils.clip_grad_norm_(params, float(max_norm))
    from pfrl_amd import _native
    from pfrl_amd.ops import _stream

    n = len(grads)
    chunks = sum((g.numel() + 4095) // 4096 for g in grads)
    dev = grads[0].device
    ws = torch.empty(chunks, dtype=torch.float64, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    G = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
    L = (ctypes.c_int64 * n)(*[g.numel() for g in grads])
    _native.check(_native.lib().pfrl_clip_grad_norm(n, G, L, float(max_norm), ws.data_ptr(),
                                                    out.data_ptr(), _stream()), "clip_grad_norm")
    return out[0]
