"""Sample + log-probability of a tanh-squashed diagonal Gaussian in one launch.

The SAC update calls ``distrib.rsample()`` and ``distrib.log_prob(actions)`` on the policy
output (``pfrl/agents/soft_actor_critic.py:228-229, 282-283``), a
``TransformedDistribution(Independent(Normal(loc, scale), 1), [TanhTransform(cache_size=1)])``
built by the example's policy head (``examples/mujoco/reproduction/soft_actor_critic/
train_soft_actor_critic.py:128-141``).  Through ``torch.distributions`` that is ~23 elementwise
launches forward and ~45 backward on [B, action_size] tensors, each ~4 us inside a captured
graph; ``sample_with_log_prob`` runs the same arithmetic as one kernel each way
(``csrc/actor.hip``) when the distribution has exactly that structure and lives on the GPU, and
calls the distribution's own methods otherwise.

The standard-normal draw is the one ``Normal.rsample`` makes (``_standard_normal`` on the
device generator), so the device random stream is consumed as before.
"""
import ctypes

import torch
from torch import distributions as D
from torch.distributions.utils import _standard_normal

from pfrl_amd import _native
from pfrl_amd._native import check


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def squashed_gaussian_params(distrib):
    """(loc, scale) when ``distrib`` is the tanh-squashed diagonal Gaussian the kernels cover,
    else None."""
    if type(distrib) is not D.TransformedDistribution:
        return None
    tr = distrib.transforms
    if len(tr) != 1 or type(tr[0]) is not D.transforms.TanhTransform or tr[0]._cache_size != 1:
        return None
    ind = distrib.base_dist
    if type(ind) is not D.Independent or ind.reinterpreted_batch_ndims != 1:
        return None
    normal = ind.base_dist
    if type(normal) is not D.Normal:
        return None
    loc, scale = normal.loc, normal.scale
    if not (loc.is_cuda and loc.dim() == 2 and loc.dtype == torch.float32
            and scale.dtype == torch.float32 and loc.shape == scale.shape and loc.shape[1] >= 1
            and loc.stride(1) == 1 and scale.stride(1) == 1 and loc.stride(0) >= loc.shape[1]
            and scale.stride(0) >= scale.shape[1] and _native.available()):
        return None
    return loc, scale


class _SquashedGaussian(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, scale, eps):
        B, A = loc.shape
        action = torch.empty((B, A), dtype=torch.float32, device=loc.device)
        both = torch.empty((2, B), dtype=torch.float32, device=loc.device)
        logp, neg = both[0], both[1]
        check(_native.lib().pfrl_squashed_gaussian_fwd(_p(loc), loc.stride(0), _p(scale), scale.stride(0),
                                                       _p(eps), _p(action), _p(logp), _p(neg), B, A,
                                                       _stream()), "squashed_gaussian_fwd")
        ctx.save_for_backward(action, eps, scale)
        ctx.mark_non_differentiable(neg)
        return action, logp, neg

    @staticmethod
    def backward(ctx, g_action, g_logp, _g_neg):
        action, eps, scale = ctx.saved_tensors
        B, A = action.shape
        g_loc = torch.empty_like(action)
        g_scale = torch.empty_like(action)
        ga = g_action.contiguous() if g_action is not None else None
        gl = g_logp.contiguous() if g_logp is not None else None
        check(_native.lib().pfrl_squashed_gaussian_bwd(_p(ga), _p(gl), _p(action), _p(eps), _p(scale),
                                                       scale.stride(0), _p(g_loc), _p(g_scale), B, A,
                                                       _stream()), "squashed_gaussian_bwd")
        return g_loc, g_scale, None


def sample_with_log_prob(distrib, reparameterize, with_negation=False):
    """``(a, distrib.log_prob(a))`` with ``a = distrib.rsample()`` (``reparameterize``) or
    ``distrib.sample()``; with ``with_negation`` a third element: ``-log_prob`` detached when the
    fused launch produced it (else None)."""
    params = squashed_gaussian_params(distrib)
    if params is None:
        a = distrib.rsample() if reparameterize else distrib.sample()
        lp = distrib.log_prob(a)
        return (a, lp, None) if with_negation else (a, lp)
    loc, scale = params
    eps = _standard_normal(loc.shape, dtype=loc.dtype, device=loc.device)
    if reparameterize:
        a, lp, neg = _SquashedGaussian.apply(loc, scale, eps)
    else:
        with torch.no_grad():
            a, lp, neg = _SquashedGaussian.apply(loc, scale, eps)
    return (a, lp, neg) if with_negation else (a, lp)
