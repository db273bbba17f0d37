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


_TORCH_STANDARD_NORMAL = _standard_normal


def _eps(shape, dtype, device):
    """The standard-normal draw of ``Normal.rsample``: through the noise feed of a captured update
    when one is installed (nn/noisy_linear.py: one launch draws every normal of the update, bit for
    bit and with the generator advanced as the separate calls would) -- unless somebody replaced
    this module's ``_standard_normal`` (tests that switch the sampling noise off)."""
    from pfrl_amd.nn import noisy_linear as nl

    feed = nl._FEED[0]
    if (feed is not None and _standard_normal is _TORCH_STANDARD_NORMAL and dtype == torch.float32
            and torch.device(device).type == "cuda"):
        n = 1
        for d in shape:
            n *= int(d)
        return feed.take_shaped(n, tuple(shape), dtype, device)
    return _standard_normal(shape, dtype=dtype, device=device)


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
        ctx.set_materialize_grads(False)    # (else a zero-fill launch per absent gradient)
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
    eps = _eps(loc.shape, loc.dtype, loc.device)
    if reparameterize:
        a, lp, neg = _SquashedGaussian.apply(loc, scale, eps)
    else:
        with torch.no_grad():
            a, lp, neg = _SquashedGaussian.apply(loc, scale, eps)
    return (a, lp, neg) if with_negation else (a, lp)


# -------------------------------------------------------------------------------------------------
# the example's head function folded into the same two launches
# -------------------------------------------------------------------------------------------------
class HeadSpec:
    """What :func:`recognise_head` found: the clamp bounds of the log-scale, the scale formula
    (0: ``sqrt(exp(2 c))``, 1: ``exp(c)``) and the action width."""

    __slots__ = ("lo", "hi", "mode", "A", "bit_exact")

    def __init__(self, lo, hi, mode, A, bit_exact):
        self.lo, self.hi, self.mode, self.A, self.bit_exact = lo, hi, mode, A, bit_exact

    def __repr__(self):
        return "HeadSpec(clamp=[%g, %g], scale=%s, A=%d, bit_exact=%s)" % (
            self.lo, self.hi, ("sqrt(exp(2c))", "exp(c)")[self.mode], self.A, self.bit_exact)


class _SquashedHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, spec):
        B, A = x.shape[0], spec.A
        action = torch.empty((B, A), dtype=torch.float32, device=x.device)
        both = torch.empty((2, B), dtype=torch.float32, device=x.device)
        logp, neg = both[0], both[1]
        check(_native.lib().pfrl_squashed_head_fwd(_p(x), x.stride(0), spec.lo, spec.hi, spec.mode,
                                                   _p(eps), _p(action), _p(logp), _p(neg), B, A,
                                                   _stream()), "squashed_head_fwd")
        ctx.save_for_backward(action, eps, x)
        ctx.spec = spec
        ctx.mark_non_differentiable(neg)
        ctx.set_materialize_grads(False)
        return action, logp, neg

    @staticmethod
    def backward(ctx, g_action, g_logp, _g_neg):
        action, eps, x = ctx.saved_tensors
        spec = ctx.spec
        B, A = action.shape
        g_x = torch.empty((B, 2 * A), dtype=torch.float32, device=x.device)
        ga = g_action.contiguous() if g_action is not None else None
        gl = g_logp.contiguous() if g_logp is not None else None
        check(_native.lib().pfrl_squashed_head_bwd(_p(ga), _p(gl), _p(action), _p(eps), _p(x),
                                                   x.stride(0), spec.lo, spec.hi, spec.mode, _p(g_x),
                                                   B, A, _stream()), "squashed_head_bwd")
        return g_x, None, None


def head_sample_with_log_prob(x, spec, reparameterize):
    """(action, log_prob, -log_prob) of the recognised head applied to ``x`` [B, 2A]: what
    ``sample_with_log_prob(head(x), reparameterize, with_negation=True)`` returns, the same draw
    from the device generator included."""
    assert x.is_cuda and x.dim() == 2 and x.shape[1] == 2 * spec.A and x.dtype == torch.float32
    if x.stride(1) != 1:
        x = x.contiguous()
    eps = _eps((x.shape[0], spec.A), x.dtype, x.device)
    if reparameterize:
        return _SquashedHead.apply(x, eps, spec)
    with torch.no_grad():
        return _SquashedHead.apply(x, eps, spec)


def _scale_formula(ls, lo, hi, mode):
    c = torch.clamp(ls, lo, hi)
    return torch.sqrt(torch.exp(c * 2)) if mode == 0 else torch.exp(c)


def recognise_head(fn, width, device):
    """Is ``fn`` (the function inside the policy's last ``Lambda``) the example's head --
    ``mean, log_scale = chunk(x, 2, dim=1)``; a tanh-squashed diagonal Gaussian with ``loc = mean``
    and ``scale = sqrt(exp(2 clamp(log_scale, lo, hi)))`` or ``exp(clamp(log_scale, lo, hi))`` --
    and if so with which bounds?  Decided by probing it on the device, the way
    ``recognise_phi`` decides about observation scalers: the returned distribution must have that
    structure, its ``loc`` must BE the first half of the input, its ``scale`` must equal one of
    the two formulas bit for bit on 4 099 log-scales (a sweep of [-30, 10], far outliers, NaN-free
    random rows), and sample, log-probability and input gradient of the fused launches must agree
    with ``fn`` + the distribution kernels on those rows.  Anything else: None, and the caller
    keeps calling ``fn``.  Consumes nothing from the global generators."""
    device = torch.device(device)
    if device.type != "cuda" or width < 2 or width % 2 or not _native.available():
        return None
    A = width // 2
    try:
        gen = torch.Generator(device=device)
        gen.manual_seed(20240924)
        sweep = torch.linspace(-30.0, 10.0, 2048, device=device)
        rnd = torch.randn(2048, generator=gen, device=device) * 4.0
        far = torch.tensor([-1e4, 1e4, 0.0], device=device)
        ls_col = torch.cat([far, sweep, rnd])
        R = ls_col.numel()
        # every action column sees every probe value (rolled, so that columns are not copies)
        ls = torch.stack([torch.roll(ls_col[3:], 7 * j) for j in range(A)], dim=1)
        ls = torch.cat([ls_col[:3, None].expand(3, A), ls], dim=0).contiguous()
        mean = torch.randn((R, A), generator=gen, device=device) * 2.0
        x = torch.cat([mean, ls], dim=1).contiguous()
        with torch.no_grad():
            d = fn(x)
        params = squashed_gaussian_params(d)
        if params is None:
            return None
        loc, scale = params
        if loc.shape != (R, A) or not torch.equal(loc, mean):
            return None
        s_lo, s_hi = scale[0], scale[1]
        if not (bool((s_lo == s_lo[0]).all()) and bool((s_hi == s_hi[0]).all())):
            return None
        found = None
        for mode in (0, 1):
            bounds = []
            for s_edge in (float(s_lo[0]), float(s_hi[0])):
                if not (s_edge > 0.0) or s_edge == float("inf"):
                    break
                import math

                guess = math.log(s_edge)
                hit = None
                for digits in range(0, 7):
                    c = round(guess, digits)
                    t = torch.full((1,), c, dtype=torch.float32, device=device)
                    if float(_scale_formula(t, c, c, mode)) == s_edge:
                        hit = c
                        break
                if hit is None:
                    break
                bounds.append(hit)
            if len(bounds) == 2 and bounds[0] <= bounds[1] and torch.equal(
                    _scale_formula(ls, bounds[0], bounds[1], mode), scale):
                found = (bounds[0], bounds[1], mode)
                break
        if found is None:
            return None
        spec = HeadSpec(found[0], found[1], found[2], A, False)
        # the fused launches against fn + the distribution launches, same eps, same upstream grads
        eps = torch.randn((R, A), generator=gen, device=device)
        g_a = torch.randn((R, A), generator=gen, device=device)
        g_l = torch.randn((R,), generator=gen, device=device)
        xa = x.clone().requires_grad_(True)
        a1, lp1, _ = _SquashedHead.apply(xa, eps, spec)
        (gx1,) = torch.autograd.grad([a1, lp1], [xa], [g_a, g_l])
        xb = x.clone().requires_grad_(True)
        loc2, scale2 = squashed_gaussian_params(fn(xb))
        a2, lp2, _ = _SquashedGaussian.apply(loc2, scale2, eps)
        (gx2,) = torch.autograd.grad([a2, lp2], [xb], [g_a, g_l])
        if not (torch.allclose(a1, a2, rtol=1e-6, atol=1e-7)
                and torch.allclose(lp1, lp2, rtol=1e-6, atol=1e-5)
                and torch.allclose(gx1, gx2, rtol=1e-5, atol=1e-6 * float(gx2.abs().max()))):
            return None
        spec.bit_exact = bool(torch.equal(a1, a2) and torch.equal(lp1, lp2) and torch.equal(gx1, gx2))
        return spec
    except Exception:      # fn does not take such an input, returns something else, ...: not the head
        return None
