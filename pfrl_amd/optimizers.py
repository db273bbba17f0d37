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


class FusedRMSprop(torch.optim.RMSprop):
    def _fusable(self, group):
        return (group["momentum"] == 0 and not group.get("maximize", False)
                and not group.get("differentiable", False))

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
            # The update is elementwise, so any dense layout works as long as the
            # parameter, its gradient and its state share it (channels_last conv
            # weights are dense permutations: the kernel walks storage order).
            ok = self._fusable(group) and all(
                p.is_cuda and p.dtype == torch.float32 and _dense(p) and not p.grad.is_sparse
                and p.grad.dtype == torch.float32 and p.grad.stride() == p.stride()
                for p in params)
            if not ok:
                return super().step(closure=None) if loss is None else loss
            centered = bool(group["centered"])
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device)
                                  if group.get("capturable", False) else torch.tensor(0.0))
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if centered:
                        st["grad_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["square_avg"].stride() != p.stride():
                    return super().step(closure=None) if loss is None else loss
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
