"""Small glue modules (reference pfrl/nn/concat_obs_and_action.py,
pfrl/nn/lmbda.py)."""
import torch
from torch import nn


class ConcatObsAndAction(nn.Module):
    """(obs, action) -> concat along the last axis (batch-flattened)."""

    def forward(self, x):
        obs, action = x
        return torch.cat([obs.reshape(obs.shape[0], -1), action.reshape(action.shape[0], -1)],
                         dim=-1)


class Lambda(nn.Module):
    def __init__(self, lambd):
        super().__init__()
        self.lambd = lambd

    def forward(self, x):
        return self.lambd(x)


def bound_by_tanh(x, low, high):
    """tanh-squash ``x`` into [low, high] (reference pfrl/functions/bound_by_tanh.py)."""
    assert isinstance(x, torch.Tensor) and low is not None and high is not None
    low = torch.as_tensor(low, dtype=x.dtype, device=x.device)
    high = torch.as_tensor(high, dtype=x.dtype, device=x.device)
    return torch.tanh(x) * ((high - low) / 2) + (high + low) / 2


class BoundByTanh(nn.Module):
    """``bound_by_tanh`` as a module.  The bounds are uploaded once per (device,
    dtype) instead of at every call (a host->device copy per forward pass, which
    also cannot be captured in a HIP graph); same arithmetic."""

    def __init__(self, low, high):
        super().__init__()
        assert low is not None and high is not None
        self.low, self.high = low, high
        self._consts = {}

    def forward(self, x):
        key = (x.device, x.dtype)
        c = self._consts.get(key)
        if c is None:
            low = torch.as_tensor(self.low, dtype=x.dtype, device=x.device)
            high = torch.as_tensor(self.high, dtype=x.dtype, device=x.device)
            c = self._consts[key] = ((high - low) / 2, (high + low) / 2)
        return torch.tanh(x) * c[0] + c[1]
