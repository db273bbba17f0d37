"""Small glue modules (reference pfrl/nn/concat_obs_and_action.py,
pfrl/nn/lmbda.py)."""
import torch
from torch import nn


class ConcatObsAndAction(nn.Module):
    """(obs, action) -> concat along the last axis (batch-flattened)."""

    def forward(self, obs_and_action):
        obs, action = obs_and_action
        return torch.cat([obs.reshape(obs.shape[0], -1), action.reshape(action.shape[0], -1)],
                         dim=-1)


class Lambda(nn.Module):
    def __init__(self, lambd):
        super().__init__()
        self.lambd = lambd

    def forward(self, x):
        return self.lambd(x)
