"""Target-network synchronisation (reference pfrl/utils/copy_param.py:4-41).

Hard sync is ``load_state_dict``; soft sync is theta' <- (1-tau) theta' + tau
theta over parameters, with BatchNorm running statistics hard-copied."""
import torch


def copy_param(target_link, source_link):
    target_link.load_state_dict(source_link.state_dict())


def _same_dense_layout(dst, src):
    return (dst.stride() == src.stride()
            and (dst.is_contiguous()
                 or (dst.dim() == 4 and dst.is_contiguous(memory_format=torch.channels_last))))


def soft_copy_params(pairs, tau):
    """Soft update of several (target, source) module pairs; on the GPU one launch for all
    their float32 tensors (pfrl_soft_update), with the per-element arithmetic of the loop
    below: dst * (1 - tau), tau * src, their sum, each rounded to f32."""
    fdst, fsrc = [], []
    for target_link, source_link in pairs:
        tgt = target_link.state_dict()
        for name, src in source_link.state_dict().items():
            dst = tgt[name]
            if dst.dtype in (torch.int32, torch.int64):
                dst.copy_(src)  # e.g. BatchNorm.num_batches_tracked
            else:
                assert dst.shape == src.shape, name     # no silent broadcasting (reference :16)
                fdst.append(dst)
                fsrc.append(src)
    if not fdst:
        return
    rest = []
    if fdst[0].is_cuda:
        import ctypes

        from pfrl_amd import _native

        fused = [(d, s) for d, s in zip(fdst, fsrc)
                 if d.is_cuda and s.is_cuda and d.device == s.device and d.dtype == torch.float32
                 and s.dtype == torch.float32 and _same_dense_layout(d, s)]
        fused_ids = {id(d) for d, _ in fused}
        rest = [(d, s) for d, s in zip(fdst, fsrc) if id(d) not in fused_ids]
        if fused:
            n = len(fused)
            Dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in fused])
            Sp = (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in fused])
            L = (ctypes.c_int64 * n)(*[d.numel() for d, _ in fused])
            stream = ctypes.c_void_p(torch.cuda.current_stream(fused[0][0].device).cuda_stream)
            _native.check(_native.lib().pfrl_soft_update(n, Dp, Sp, L, float(tau), stream),
                          "soft_update")
    else:
        rest = list(zip(fdst, fsrc))
    for dst, src in rest:
        dst.mul_(1 - tau)
        dst.add_(tau * src)


def soft_copy_param(target_link, source_link, tau):
    soft_copy_params([(target_link, source_link)], tau)


def copy_grad(target_link, source_link):
    for tp, sp in zip(target_link.parameters(), source_link.parameters()):
        assert tp.shape == sp.shape
        tp.grad = None if sp.grad is None else sp.grad.clone()


def synchronize_parameters(src, dst, method, tau=None):
    if method == "hard":
        copy_param(dst, src)
    elif method == "soft":
        soft_copy_param(dst, src, tau)
    else:
        raise ValueError("unknown target update method %r" % (method,))
