"""Factorised-Gaussian NoisyNet linear layers (http://arxiv.org/abs/1706.10295).

A noisy layer keeps two ordinary linear layers, ``mu`` and ``sigma``; every
forward pass draws in + out unit Gaussians r, shapes them with
f(r) = sign(r) sqrt(|r|), and uses

    W = mu.W + sigma.W * outer(f(r_out), f(r_in)),   b = mu.b + sigma.b * f(r_out)

Initialisation follows the reference (pfrl/nn/noisy_linear.py): mu.W uniform
with the LeCun bound scaled by 1/sqrt(3), sigma.* constant sigma_scale /
sqrt(fan_in).  The noise comes from the torch generator of the layer's device.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _fan_in(tensor):
    return tensor.shape[0] if tensor.ndim == 1 else \
        torch.nn.init._calculate_correct_fan(tensor, "fan_in")


@torch.no_grad()
def init_lecun_uniform(tensor, scale=1.0):
    bound = scale * math.sqrt(3.0 / torch.nn.init._calculate_correct_fan(tensor, "fan_in"))
    return tensor.uniform_(-bound, bound)


@torch.no_grad()
def init_variance_scaling_constant(tensor, scale=1.0):
    return tensor.fill_(scale / math.sqrt(_fan_in(tensor)))


# Where the layers' Gaussian draws come from.  None: every forward calls torch.randn (the
# reference's torch.normal, same kernel, same stream).  A captured update installs a feed
# (agents/graphed_update.py): first a RECORDING one -- the layers draw as usual and the sizes of
# their draws are noted in call order -- then a SERVING one whose take() hands out views of one
# static buffer in the same order; that buffer is filled before every replay by ONE launch that
# reproduces those torch.randn calls bit for bit and advances the generator as they would
# (ops.randn_calls, csrc/philox.hip).
_FEED = [None]


class NoiseFeed:
    def __init__(self, views=None):
        self.sizes = []          # recording: the draws seen so far
        self.views = views       # serving: one tensor per draw, in call order
        self.at = 0

    def take(self, n, like):
        if self.views is None:
            self.sizes.append(int(n))
            return torch.randn(n, dtype=like.dtype, device=like.device)
        # (no wrap-around: a step that draws MORE normals than the recording warm-up would be
        # served the same noise twice, silently -- ADVICE r5)
        assert self.at < len(self.views), "noise feed exhausted: the step draws more than was recorded"
        v = self.views[self.at]
        self.at += 1
        assert v.numel() == n and v.device == like.device, "noise feed out of step with the layers"
        return v

    def take_shaped(self, n, shape, dtype, device):
        """The same for a draw of another shape (``torch.empty(shape).normal_()`` of
        ``Normal.rsample``: the kernel torch.randn(n) launches)."""
        if self.views is None:
            self.sizes.append(int(n))
            return torch.empty(shape, dtype=dtype, device=device).normal_()
        assert self.at < len(self.views), "noise feed exhausted: the step draws more than was recorded"
        v = self.views[self.at]
        self.at += 1
        assert v.numel() == n, "noise feed out of step with the samplers"
        return v.view(shape)


class noise_feed:
    """``with noise_feed(feed):`` -- the layers draw through ``feed`` (None: torch.randn)."""

    def __init__(self, feed):
        self.feed = feed

    def __enter__(self):
        self.saved, _FEED[0] = _FEED[0], self.feed
        return self.feed

    def __exit__(self, *exc):
        _FEED[0] = self.saved


def _draw(n, like):
    feed = _FEED[0]
    if feed is not None and like.is_cuda and like.dtype == torch.float32:
        return feed.take(n, like)
    return torch.randn(n, dtype=like.dtype, device=like.device)


def _shaped_noise(n, like):
    r = _draw(n, like)
    return r.sign() * r.abs().sqrt()


class FactorizedNoisyLinear(nn.Module):
    def __init__(self, mu_link, sigma_scale=0.4):
        super().__init__()
        out_features, in_features = mu_link.weight.shape
        self.out_size = out_features
        self.hasbias = mu_link.bias is not None
        dev = mu_link.weight.device
        self.mu = nn.Linear(in_features, out_features, bias=self.hasbias).to(dev)
        self.sigma = nn.Linear(in_features, out_features, bias=self.hasbias).to(dev)
        init_lecun_uniform(self.mu.weight, scale=1 / math.sqrt(3))
        init_variance_scaling_constant(self.sigma.weight, scale=sigma_scale)
        if self.hasbias:
            init_variance_scaling_constant(self.sigma.bias, scale=sigma_scale)

    def forward(self, x, relu=False):
        """``relu=True`` (used by ``linear_activation``): ReLU applied to the result, inside the
        GEMM's epilogue where the MFMA linear kernels take the layer."""
        y = self._forward(x, relu)
        return y

    def _forward(self, x, relu):
        out_features, in_features = self.sigma.weight.shape
        sw = self.sigma.weight
        if sw.is_cuda:
            from pfrl_amd import ops

            if ops.noisy_weights_supported(sw) and self.mu.weight.is_contiguous():
                # same draw as below (one normal_ of in + out values); shaping, outer
                # product and both addcmul fused into one launch (and one for backward)
                r = _draw(in_features + out_features, sw)
                from pfrl_amd.nn import mfma_linear

                if self.hasbias and mfma_linear.noisy_supported(x, self.mu.weight, sw, self.mu.bias,
                                                                self.sigma.bias):
                    # the perturbed weights never exist: formed in the GEMM's operand loader
                    return mfma_linear._NoisyLinear.apply(x, self.mu.weight, sw, self.mu.bias,
                                                          self.sigma.bias, r, bool(relu))
                weight, bias = ops.noisy_weights(
                    self.mu.weight, sw, self.mu.bias if self.hasbias else None,
                    self.sigma.bias if self.hasbias else None, r)
                from pfrl_amd.nn import mfma_linear

                if (bias is not None and weight.is_contiguous()
                        and mfma_linear.supported_tensors(x, weight, bias)):
                    # the noisy weights are a dense [out, in] matrix: the GEMM (and its
                    # backward towards mu / sigma) on the MFMA linear kernels
                    return mfma_linear._Linear.apply(
                        x, weight, bias, bool(relu),
                        mfma_linear.noisy_fwd_splits(x.shape[0], weight.shape[0], weight.shape[1]))
                y = F.linear(x, weight, bias)
                return F.relu(y) if relu else y
        noise = _shaped_noise(in_features + out_features, self.sigma.weight)
        eps_in, eps_out = noise[:in_features], noise[in_features:]
        weight = torch.addcmul(self.mu.weight, self.sigma.weight, torch.outer(eps_out, eps_in))
        bias = torch.addcmul(self.mu.bias, self.sigma.bias, eps_out) if self.hasbias else None
        y = F.linear(x, weight, bias)
        return F.relu(y) if relu else y


def to_factorized_noisy(module, *args, **kwargs):
    """Swap every nn.Linear below ``module`` for a FactorizedNoisyLinear
    (reference pfrl/nn/noisy_chain.py)."""
    for name, child in list(module.named_children()):
        if isinstance(child, nn.Linear):
            setattr(module, name, FactorizedNoisyLinear(child, *args, **kwargs))
        else:
            to_factorized_noisy(child, *args, **kwargs)
