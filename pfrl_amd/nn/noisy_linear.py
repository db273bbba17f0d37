"""Factorised NoisyNet layers (http://arxiv.org/abs/1706.10295; reference
pfrl/nn/noisy_linear.py:25-70, pfrl/nn/noisy_chain.py).  The noise comes from
the torch generator of the layer's device; stock PyTorch."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def init_lecun_uniform(tensor, scale=1.0):
    fan_in = torch.nn.init._calculate_correct_fan(tensor, "fan_in")
    s = scale * np.sqrt(3.0 / fan_in)
    with torch.no_grad():
        return tensor.uniform_(-s, s)


def init_variance_scaling_constant(tensor, scale=1.0):
    if tensor.ndim == 1:
        s = scale / np.sqrt(tensor.shape[0])
    else:
        s = scale / np.sqrt(torch.nn.init._calculate_correct_fan(tensor, "fan_in"))
    with torch.no_grad():
        return tensor.fill_(s)


class FactorizedNoisyLinear(nn.Module):
    """W = mu_W + sigma_W * (eps_out eps_in^T), b = mu_b + sigma_b * eps_out with
    eps = sign(r) sqrt(|r|), r ~ N(0, 1), redrawn on every forward."""

    def __init__(self, mu_link, sigma_scale=0.4):
        super().__init__()
        self.out_size = mu_link.out_features
        self.hasbias = mu_link.bias is not None
        in_size = mu_link.weight.shape[1]
        device = mu_link.weight.device
        self.mu = nn.Linear(in_size, self.out_size, bias=self.hasbias)
        init_lecun_uniform(self.mu.weight, scale=1 / np.sqrt(3))
        self.sigma = nn.Linear(in_size, self.out_size, bias=self.hasbias)
        init_variance_scaling_constant(self.sigma.weight, scale=sigma_scale)
        if self.hasbias:
            init_variance_scaling_constant(self.sigma.bias, scale=sigma_scale)
        self.mu.to(device)
        self.sigma.to(device)

    def _eps(self, n, dtype, device):
        r = torch.normal(mean=0.0, std=1.0, size=(n,), dtype=dtype, device=device)
        return torch.abs(torch.sqrt(torch.abs(r))) * torch.sign(r)

    def forward(self, x):
        dtype = self.sigma.weight.dtype
        out_size, in_size = self.sigma.weight.shape
        eps = self._eps(in_size + out_size, dtype, self.sigma.weight.device)
        eps_x, eps_y = eps[:in_size], eps[in_size:]
        W = torch.addcmul(self.mu.weight, self.sigma.weight, torch.outer(eps_y, eps_x))
        if self.hasbias:
            return F.linear(x, W, torch.addcmul(self.mu.bias, self.sigma.bias, eps_y))
        return F.linear(x, W)


def to_factorized_noisy(module, *args, **kwargs):
    """Replace every nn.Linear below ``module`` by a FactorizedNoisyLinear."""
    for name, child in module.named_children():
        if isinstance(child, nn.Linear):
            module._modules[name] = FactorizedNoisyLinear(child, *args, **kwargs)
        else:
            to_factorized_noisy(child, *args, **kwargs)
