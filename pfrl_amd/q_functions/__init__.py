"""Q-function heads (reference pfrl/q_functions/state_q_functions.py:27-77,
pfrl/q_functions/__init__.py DiscreteActionValueHead)."""
import torch.nn as nn
import torch.nn.functional as F

from pfrl_amd.action_value import DiscreteActionValue
from pfrl_amd.nn.mlp import MLP


class DiscreteActionValueHead(nn.Module):
    def forward(self, q_values):
        return DiscreteActionValue(q_values)


class SingleModelStateQFunctionWithDiscreteAction(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, x):
        return DiscreteActionValue(self.model(x))


class FCStateQFunctionWithDiscreteAction(SingleModelStateQFunctionWithDiscreteAction):
    def __init__(self, ndim_obs, n_actions, n_hidden_channels, n_hidden_layers,
                 nonlinearity=F.relu, last_wscale=1.0):
        super().__init__(model=MLP(in_size=ndim_obs, out_size=n_actions,
                                   hidden_sizes=[n_hidden_channels] * n_hidden_layers,
                                   nonlinearity=nonlinearity, last_wscale=last_wscale))
