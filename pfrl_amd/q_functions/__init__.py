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


def scale_by_tanh(x, low, high):
    """tanh(x) mapped affinely onto the box [low, high] (NumPy bounds, unbatched)."""
    import numpy as _np
    import torch as _t

    low, high = _np.asarray(low), _np.asarray(high)
    half_width = _t.from_numpy((high - low) / 2).unsqueeze(0).to(x.device)
    centre = _t.from_numpy((high + low) / 2).unsqueeze(0).to(x.device)
    return _t.tanh(x) * half_width + centre


class FCQuadraticStateQFunction(nn.Module):
    """Fully connected NAF Q-function for continuous actions (http://arxiv.org/abs/1603.00748;
    reference state_q_functions.py:143-219): a ReLU trunk feeding V(s), mu(s) (squashed onto the
    action box if ``scale_mu``) and a lower-triangular factor L(s) with exponentiated diagonal;
    the advantage matrix is ``L L^T``.  Returns a ``QuadraticActionValue``."""

    def __init__(self, n_input_channels, n_dim_action, n_hidden_channels, n_hidden_layers,
                 action_space, scale_mu=True):
        assert action_space is not None and n_hidden_layers >= 1
        super().__init__()
        from pfrl_amd.initializers import init_chainer_default

        self.n_input_channels, self.n_dim_action = n_input_channels, n_dim_action
        self.n_hidden_channels, self.n_hidden_layers = n_hidden_channels, n_hidden_layers
        self.action_space, self.scale_mu = action_space, scale_mu

        def linear(n_in, n_out):
            return init_chainer_default(nn.Linear(n_in, n_out))

        widths = [n_input_channels] + [n_hidden_channels] * n_hidden_layers
        self.hidden_layers = nn.ModuleList(linear(a, b) for a, b in zip(widths, widths[1:]))
        self.v = linear(n_hidden_channels, 1)
        self.mu = linear(n_hidden_channels, n_dim_action)
        self.mat_diag = linear(n_hidden_channels, n_dim_action)
        n_below = n_dim_action * (n_dim_action - 1) // 2
        if n_below > 0:
            self.mat_non_diag = linear(n_hidden_channels, n_below)

    def forward(self, state):
        import torch as _t

        from pfrl_amd.action_value import QuadraticActionValue
        from pfrl_amd.functions.lower_triangular_matrix import lower_triangular_matrix

        h = state
        for layer in self.hidden_layers:
            h = F.relu(layer(h))
        mu = self.mu(h)
        if self.scale_mu:
            mu = scale_by_tanh(mu, high=self.action_space.high, low=self.action_space.low)
        diag = _t.exp(self.mat_diag(h))
        if hasattr(self, "mat_non_diag"):
            factor = lower_triangular_matrix(diag, self.mat_non_diag(h))
            mat = _t.matmul(factor, factor.transpose(1, 2))
        else:
            mat = (diag ** 2).unsqueeze(2)
        return QuadraticActionValue(mu, mat, self.v(h), min_action=self.action_space.low,
                                    max_action=self.action_space.high)


from pfrl_amd.q_functions.dueling_dqn import DistributionalDuelingDQN, DuelingDQN  # NOQA,E402
