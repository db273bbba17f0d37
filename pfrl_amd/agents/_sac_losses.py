"""The three elementwise chains of the SAC update as single launches on the GPU.

``soft_target_q``, ``half_mse`` and ``policy_loss`` state the formulas of
``pfrl/agents/soft_actor_critic.py:226-248, 284-291`` once in PyTorch operations (the route
for CPU tensors and anything the kernels do not cover) and once through ``csrc/actor.hip``
(f32 [B] vectors on the GPU): ~9 + 2 x 7 + ~20 launches of 256 elements become 1 + 2 x 2 + 2.
``temperature`` is either the fixed float or the ``TemperatureHolder``'s ``log_temperature``
parameter, whose ``exp`` the kernels take themselves.
"""
import ctypes

import torch
import torch.nn.functional as F

from pfrl_amd import _native
from pfrl_amd._native import check


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _vec_ok(*ts):
    return (_native.available() and all(
        torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        for t in ts) and ts[0].numel() >= 1 and all(t.numel() == ts[0].numel() for t in ts))


def _temperature_args(temperature):
    """(device pointer to log T or None, float T) from a float or a log-temperature tensor."""
    if torch.is_tensor(temperature):
        return _p(temperature), 0.0
    return None, float(temperature)


def _temperature_value(temperature):
    return torch.exp(temperature.detach()) if torch.is_tensor(temperature) else temperature


_UNIT = {}


def unit_grad(device):
    """THE scalar 1 that ``loss.backward(unit_grad(device))`` passes as dL/dloss.  The loss
    nodes below recognise it by address: their forward launch has then already written the
    gradients (``unit_g_*`` of the C ABI) and backward launches nothing."""
    device = torch.device(device)
    one = _UNIT.get(device)
    if one is None:
        one = _UNIT[device] = torch.ones((), dtype=torch.float32, device=device)
    return one


def _is_unit(g):
    one = _UNIT.get(g.device) if g is not None else None
    return one is not None and g.data_ptr() == one.data_ptr()


def soft_target_q(reward, discount, terminal, next_q1, next_q2, next_log_prob, temperature):
    """reward + discount * (1 - terminal) * (min(next_q1, next_q2) - T * next_log_prob), no
    gradient.  ``temperature``: float, or the scalar log-temperature tensor (T = exp of it)."""
    with torch.no_grad():
        if (_vec_ok(reward, discount, terminal, next_q1, next_q2, next_log_prob)
                and (not torch.is_tensor(temperature) or temperature.is_cuda)):
            B = reward.numel()
            out = torch.empty((B,), dtype=torch.float32, device=reward.device)
            lt, tv = _temperature_args(temperature)
            check(_native.lib().pfrl_sac_target_q(_p(reward), _p(discount), _p(terminal), _p(next_q1),
                                                  _p(next_q2), _p(next_log_prob), lt, tv, _p(out), B,
                                                  _stream()), "sac_target_q")
            return out
        next_q = torch.min(next_q1, next_q2)
        entropy_term = _temperature_value(temperature) * next_log_prob[..., None]
        assert next_q.shape == entropy_term.shape
        return reward + discount * (1.0 - terminal) * torch.flatten(next_q - entropy_term)


class _HalfMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, pred):
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        check(_native.lib().pfrl_half_mse_fwd(_p(target), _p(pred), _p(loss), pred.numel(), _stream()),
              "half_mse_fwd")
        ctx.save_for_backward(target, pred)
        return loss

    @staticmethod
    def backward(ctx, g):
        target, pred = ctx.saved_tensors
        g_pred = torch.empty_like(pred)
        check(_native.lib().pfrl_half_mse_bwd(_p(g.contiguous()), _p(target), _p(pred), _p(g_pred),
                                              pred.numel(), _stream()), "half_mse_bwd")
        return None, g_pred


def half_mse(target, pred):
    """0.5 * F.mse_loss(target, pred); ``target`` carries no gradient."""
    if _vec_ok(target, pred) and not target.requires_grad:
        return _HalfMse.apply(target, pred)
    return 0.5 * F.mse_loss(target, pred)


class _HalfMsePair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, pred1, pred2):
        loss = torch.empty((2,), dtype=torch.float32, device=target.device)
        unit = torch.empty((2, target.numel()), dtype=torch.float32, device=target.device)
        P = (ctypes.c_void_p * 2)(pred1.data_ptr(), pred2.data_ptr())
        Lp = (ctypes.c_void_p * 2)(loss[0].data_ptr(), loss[1].data_ptr())
        U = (ctypes.c_void_p * 2)(unit[0].data_ptr(), unit[1].data_ptr())
        check(_native.lib().pfrl_half_mse_twin_fwd(_p(target), P, Lp, U, target.numel(), _stream()),
              "half_mse_twin_fwd")
        ctx.save_for_backward(target, pred1, pred2, unit)
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g1, g2):
        target, pred1, pred2, unit = ctx.saved_tensors
        if _is_unit(g1) and _is_unit(g2):       # written by the forward launch
            return None, unit[0].view(pred1.shape), unit[1].view(pred2.shape)
        gp = torch.empty((2, target.numel()), dtype=torch.float32, device=target.device)
        g1 = g1.contiguous() if g1 is not None else None
        g2 = g2.contiguous() if g2 is not None else None
        G = (ctypes.c_void_p * 2)(g1.data_ptr() if g1 is not None else 0,
                                  g2.data_ptr() if g2 is not None else 0)
        P = (ctypes.c_void_p * 2)(pred1.data_ptr(), pred2.data_ptr())
        O = (ctypes.c_void_p * 2)(gp[0].data_ptr(), gp[1].data_ptr())
        check(_native.lib().pfrl_half_mse_twin_bwd(G, _p(target), P, O, target.numel(), _stream()),
              "half_mse_twin_bwd")
        return None, gp[0].view(pred1.shape), gp[1].view(pred2.shape)


def half_mse_pair(target, pred1, pred2):
    """(half_mse(target, pred1), half_mse(target, pred2)), one launch each way on the GPU."""
    if _vec_ok(target, pred1, pred2) and not target.requires_grad:
        return _HalfMsePair.apply(target, pred1, pred2)
    return half_mse(target, pred1), half_mse(target, pred2)


class _PolicyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_prob, q1, q2, temperature):
        loss = torch.empty((), dtype=torch.float32, device=q1.device)
        lt, tv = _temperature_args(temperature)
        B = q1.numel()
        unit = torch.empty((3, B), dtype=torch.float32, device=q1.device)
        check(_native.lib().pfrl_sac_policy_loss_fwd(_p(log_prob), _p(q1), _p(q2), lt, tv, _p(loss),
                                                     _p(unit[0]), _p(unit[1]), _p(unit[2]), B,
                                                     _stream()), "sac_policy_loss_fwd")
        ctx.save_for_backward(q1, q2, temperature if torch.is_tensor(temperature) else None, unit)
        ctx.t_val = tv
        ctx.lp_shape = log_prob.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        q1, q2, log_t, unit = ctx.saved_tensors
        if _is_unit(g):                          # written by the forward launch
            return unit[0].view(ctx.lp_shape), unit[1].view(q1.shape), unit[2].view(q2.shape), None
        g_lp = torch.empty(q1.numel(), dtype=torch.float32, device=q1.device)
        g1, g2 = torch.empty_like(q1), torch.empty_like(q2)
        check(_native.lib().pfrl_sac_policy_loss_bwd(_p(g.contiguous()), _p(q1), _p(q2), _p(log_t),
                                                     ctx.t_val, _p(g_lp), _p(g1), _p(g2), q1.numel(),
                                                     _stream()), "sac_policy_loss_bwd")
        return g_lp, g1, g2, None


def policy_loss(log_prob, q1, q2, temperature):
    """mean(T * log_prob[:, None] - min(q1, q2)) for log_prob [B], q1, q2 [B, 1]; the temperature
    carries no gradient here (it has its own loss)."""
    if (_vec_ok(log_prob, q1, q2) and q1.dim() == 2
            and (not torch.is_tensor(temperature) or temperature.is_cuda)):
        t = temperature.detach() if torch.is_tensor(temperature) else temperature
        return _PolicyLoss.apply(log_prob, q1, q2, t)
    q = torch.min(q1, q2)
    entropy_term = _temperature_value(temperature) * log_prob[..., None]
    assert q.shape == entropy_term.shape
    return torch.mean(entropy_term - q)


class _TemperatureLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_temperature, log_prob, entropy_target):
        loss = torch.empty((), dtype=torch.float32, device=log_prob.device)
        check(_native.lib().pfrl_sac_temperature_loss(_p(log_temperature), _p(log_prob),
                                                      float(entropy_target), _p(loss), log_prob.numel(),
                                                      _stream()), "sac_temperature_loss")
        ctx.save_for_backward(loss)
        return loss

    @staticmethod
    def backward(ctx, g):
        (loss,) = ctx.saved_tensors
        # d/dlog T of -mean(exp(log T) * c) is the loss itself
        return g * loss, None, None


def temperature_loss(temperature_holder, log_prob, entropy_target):
    """-mean(T * (log_prob + entropy_target)) with T = temperature_holder()."""
    log_t = getattr(temperature_holder, "log_temperature", None)
    if (log_t is not None and _vec_ok(log_prob) and log_t.is_cuda and log_t.dtype == torch.float32
            and log_t.numel() == 1 and not log_prob.requires_grad):
        return _TemperatureLoss.apply(log_t, log_prob, entropy_target)
    return -torch.mean(temperature_holder() * (log_prob + entropy_target))


def temperature_step(temperature_holder, log_prob, entropy_target, optimizer):
    """The temperature loss AND its optimizer step as one launch (pfrl_sac_temperature_step) when
    the optimizer is a FusedAdam over exactly the log-temperature: d loss / d log T is the loss
    itself, so no autograd pass is needed.  Returns the loss, or None when not applicable (the
    caller then computes the loss and steps as usual)."""
    from pfrl_amd.optimizers import FusedAdam

    log_t = getattr(temperature_holder, "log_temperature", None)
    if not (type(optimizer) is FusedAdam and log_t is not None and _vec_ok(log_prob) and log_t.is_cuda
            and log_t.dtype == torch.float32 and log_t.numel() == 1 and not log_prob.requires_grad
            and len(optimizer.param_groups) == 1):
        return None
    group = optimizer.param_groups[0]
    if (len(group["params"]) != 1 or group["params"][0] is not log_t or group["amsgrad"]
            or group.get("maximize", False) or isinstance(group["lr"], torch.Tensor)
            or not optimizer._prepare_state(group, [log_t])):
        return None
    st = optimizer.state[log_t]
    loss = torch.empty((), dtype=torch.float32, device=log_prob.device)
    b1, b2 = group["betas"]
    with torch.no_grad():
        check(_native.lib().pfrl_sac_temperature_step(
            _p(log_t), _p(log_prob), float(entropy_target), _p(loss), _p(st["exp_avg"]),
            _p(st["exp_avg_sq"]), _p(st["step"]), float(group["lr"]), float(b1), float(b2),
            float(group["eps"]), float(group["weight_decay"]), log_prob.numel(), _stream()),
            "sac_temperature_step")
    return loss
