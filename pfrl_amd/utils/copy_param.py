"""Target-network synchronisation (reference pfrl/utils/copy_param.py:4-41).

Hard sync is ``load_state_dict``; soft sync is theta' <- (1-tau) theta' + tau
theta over parameters, with BatchNorm running statistics hard-copied."""
import torch


def copy_param(target_link, source_link):
    target_link.load_state_dict(source_link.state_dict())


def soft_copy_param(target_link, source_link, tau):
    tgt = target_link.state_dict()
    fdst, fsrc = [], []
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
    if fdst[0].is_cuda:
        # the same three roundings per element (dst*(1-tau), tau*src, sum) as the
        # per-tensor loop, as three multi-tensor launches instead of 3 per tensor
        torch._foreach_mul_(fdst, 1 - tau)
        torch._foreach_add_(fdst, torch._foreach_mul(fsrc, tau))
        return
    for dst, src in zip(fdst, fsrc):
        dst.mul_(1 - tau)
        dst.add_(tau * src)


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
