"""Policy heads (reference pfrl/policies: softmax_policy.py,
gaussian_policy.py).  They wrap network outputs in torch.distributions and
stay stock PyTorch."""
import numpy as np
import torch
from torch import nn


class SoftmaxCategoricalHead(nn.Module):
    def forward(self, logits):
        return torch.distributions.Categorical(logits=logits)


class GaussianHeadWithStateIndependentCovariance(nn.Module):
    """Diagonal Gaussian whose (transformed) variance is a free parameter
    (reference pfrl/policies/gaussian_policy.py:49-88)."""

    def __init__(self, action_size, var_type="spherical", var_func=nn.functional.softplus,
                 var_param_init=0):
        super().__init__()
        self.var_func = var_func
        var_size = {"spherical": 1, "diagonal": action_size}[var_type]
        self.var_param = nn.Parameter(
            torch.tensor(np.broadcast_to(var_param_init, var_size), dtype=torch.float))

    def forward(self, mean):
        var = self.var_func(self.var_param)
        return torch.distributions.Independent(
            torch.distributions.Normal(loc=mean, scale=torch.sqrt(var)), 1)


class GaussianHeadWithDiagonalCovariance(nn.Module):
    """Input = concat(mean, pre-activation variance)
    (reference gaussian_policy.py:8-46)."""

    def __init__(self, var_func=nn.functional.softplus):
        super().__init__()
        self.var_func = var_func

    def forward(self, mean_and_var):
        assert mean_and_var.ndim == 2
        mean, pre_var = mean_and_var.chunk(2, dim=1)
        scale = self.var_func(pre_var).sqrt()
        return torch.distributions.Independent(
            torch.distributions.Normal(loc=mean, scale=scale), 1)


class GaussianHeadWithFixedCovariance(nn.Module):
    """Diagonal Gaussian around the network output with a constant scale (reference
    gaussian_policy.py:97-124)."""

    def __init__(self, scale=1):
        super().__init__()
        self.scale = scale

    def forward(self, mean):
        return torch.distributions.Independent(
            torch.distributions.Normal(loc=mean, scale=self.scale), 1)


class DeterministicHead(nn.Module):
    """Deterministic policy output as a distribution (reference
    pfrl/policies/deterministic_policy.py:7-11)."""

    def forward(self, loc):
        from pfrl_amd.distributions import Delta

        return torch.distributions.Independent(Delta(loc=loc), 1)
from pfrl_amd.policies import deterministic_policy, gaussian_policy, softmax_policy  # NOQA,E402
