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


class DistributionalSingleModelStateQFunctionWithDiscreteAction(nn.Module):
    """model(x) -> (batch, n_actions, n_atoms) probabilities over ``z_values``."""

    def __init__(self, model, z_values):
        super().__init__()
        import torch as _t

        self.model = model
        self.register_buffer("z_values", _t.as_tensor(z_values, dtype=_t.float32))

    def forward(self, x):
        from pfrl_amd.action_value import DistributionalDiscreteActionValue

        return DistributionalDiscreteActionValue(self.model(x), self.z_values)


class _AtomSoftmax(nn.Module):
    """(batch, n_actions * n_atoms) logits -> (batch, n_actions, n_atoms) probabilities."""

    def __init__(self, n_actions, n_atoms):
        super().__init__()
        self.n_actions, self.n_atoms = n_actions, n_atoms

    def forward(self, h):
        return F.softmax(h.reshape(-1, self.n_actions, self.n_atoms), dim=2)


class DistributionalFCStateQFunctionWithDiscreteAction(
        DistributionalSingleModelStateQFunctionWithDiscreteAction):
    """Fully connected categorical (C51) Q-function: MLP -> (n_actions, n_atoms) softmax
    over ``n_atoms`` evenly spaced on [v_min, v_max] (reference state_q_functions.py:99-145)."""

    def __init__(self, ndim_obs, n_actions, n_atoms, v_min, v_max, n_hidden_channels,
                 n_hidden_layers, nonlinearity=F.relu, last_wscale=1.0):
        assert n_atoms >= 2 and v_min < v_max
        import numpy as _np

        model = nn.Sequential(
            MLP(in_size=ndim_obs, out_size=n_actions * n_atoms,
                hidden_sizes=[n_hidden_channels] * n_hidden_layers, nonlinearity=nonlinearity,
                last_wscale=last_wscale), _AtomSoftmax(n_actions, n_atoms))
        super().__init__(model=model,
                         z_values=_np.linspace(v_min, v_max, num=n_atoms, dtype=_np.float32))


from pfrl_amd.q_functions.dueling_dqn import DistributionalDuelingDQN, DuelingDQN  # NOQA,E402
