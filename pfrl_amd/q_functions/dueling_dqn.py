"""Dueling heads (reference pfrl/q_functions/dueling_dqn.py: DuelingDQN :20-58,
DistributionalDuelingDQN :61-129).

Module and parameter names (``conv_layers.<i>``, ``a_stream``, ``v_stream``, ``main_stream``) and
the construction order -- hence the consumption of the torch RNG under a seed -- follow the
reference, so its checkpoints load strictly and seeded runs start from the same weights.

Where the device path differs from a plain forward pass (same values, fewer launches):

* each convolution's bias + ReLU runs as one fused HIP launch (``conv_activation``; the last
  one writes the flattened NCHW layout the first linear layer expects even when the trunk is
  channels-last, ``planar_out``);
* ``DistributionalDuelingDQN`` combines advantage and value logits, subtracts the advantage
  mean and takes the per-action softmax over atoms in one launch forwards and one backwards
  (``ops.dueling_softmax`` -> ``pfrl_dueling_softmax_fwd`` / ``_bwd``) on a GPU; on the CPU the
  same arithmetic is spelled out with torch ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from pfrl_amd import action_value
from pfrl_amd.initializers import init_chainer_default
from pfrl_amd.nn.atari_cnn import (constant_bias_initializer, conv_activation,
                                   linear_activation)
from pfrl_amd.nn.mlp import MLP


def _nature_convs(n_input_channels):
    return nn.ModuleList([nn.Conv2d(n_input_channels, 32, 8, stride=4),
                          nn.Conv2d(32, 64, 4, stride=2),
                          nn.Conv2d(64, 64, 3, stride=1)])


class DuelingDQN(nn.Module):
    """http://arxiv.org/abs/1511.06581"""

    def __init__(self, n_actions, n_input_channels=4, activation=F.relu, bias=0.1):
        super().__init__()
        self.n_actions = n_actions
        self.n_input_channels = n_input_channels
        self.activation = activation
        self.conv_layers = _nature_convs(n_input_channels)
        self.a_stream = MLP(3136, n_actions, [512])
        self.v_stream = MLP(3136, 1, [512])
        self.conv_layers.apply(init_chainer_default)
        self.conv_layers.apply(constant_bias_initializer(bias=bias))

    def forward(self, x):
        h = _conv_trunk(self.conv_layers, x, self.activation)
        batch_size = x.shape[0]
        h = h.reshape(batch_size, -1)
        ya = self.a_stream(h)
        mean = torch.reshape(torch.sum(ya, dim=1) / self.n_actions, (batch_size, 1))
        ya = ya - mean
        ys = self.v_stream(h)
        return action_value.DiscreteActionValue(ya + ys)


def _conv_trunk(conv_layers, x, activation):
    """activation(conv(...)) over the layers; on the GPU as the MFMA trunk kernels (one
    autograd node, bias + ReLU in the epilogues) when the shapes are inside what they cover,
    else layer by layer.  Either way the result is [N, C, H, W] in plain NCHW memory."""
    if x.is_cuda:
        from pfrl_amd.nn import mfma_trunk
        from pfrl_amd.nn.atari_cnn import _is_relu

        if _is_relu(activation):
            specs = mfma_trunk.plan_for(conv_layers, None, x)
            if specs is not None:
                return mfma_trunk.trunk_forward(x, specs, list(conv_layers), None)
    h = x
    last = len(conv_layers) - 1
    for i, layer in enumerate(conv_layers):
        h = conv_activation(layer, h, activation, planar_out=(i == last))
    return h


class DistributionalDuelingDQN(nn.Module):
    """Distributional dueling Q-function (Rainbow head)."""

    def __init__(self, n_actions, n_atoms, v_min, v_max, n_input_channels=4,
                 activation=torch.relu, bias=0.1):
        assert n_atoms >= 2
        assert v_min < v_max
        super().__init__()
        self.n_actions = n_actions
        self.n_input_channels = n_input_channels
        self.activation = activation
        self.n_atoms = n_atoms
        self.register_buffer("z_values",
                             torch.linspace(v_min, v_max, n_atoms, dtype=torch.float32),
                             persistent=False)
        self.conv_layers = _nature_convs(n_input_channels)
        self.main_stream = nn.Linear(3136, 1024)
        self.a_stream = nn.Linear(512, n_actions * n_atoms)
        self.v_stream = nn.Linear(512, n_atoms)
        self.apply(init_chainer_default)
        self.conv_layers.apply(constant_bias_initializer(bias=bias))

    def _noisy_pair(self, h):
        """(a_stream(h[:, :half]), v_stream(h[:, half:])) of two factorised NoisyNet streams as ONE
        launch that reads the halves of ``h`` in place (nn/mfma_linear.py::_NoisyLinearPair), or
        None.  The streams draw their noise in the order of the separate calls: a, then v."""
        from pfrl_amd.nn.noisy_linear import FactorizedNoisyLinear, _draw

        a, v = self.a_stream, self.v_stream
        if not (type(a) is FactorizedNoisyLinear and type(v) is FactorizedNoisyLinear and h.is_cuda):
            return None
        from pfrl_amd.nn import mfma_linear

        if not mfma_linear.noisy_pair_supported(h, a, v):
            return None
        ra = _draw(a.mu.weight.shape[1] + a.mu.weight.shape[0], a.sigma.weight)
        rv = _draw(v.mu.weight.shape[1] + v.mu.weight.shape[0], v.sigma.weight)
        return mfma_linear._NoisyLinearPair.apply(
            h, a.mu.weight, a.sigma.weight, a.mu.bias, a.sigma.bias, ra,
            v.mu.weight, v.sigma.weight, v.mu.bias, v.sigma.bias, rv)

    def forward(self, x):
        h = _conv_trunk(self.conv_layers, x, self.activation)
        batch_size = x.shape[0]
        h = linear_activation(self.main_stream, h.reshape(batch_size, -1), self.activation)
        pair = self._noisy_pair(h)
        ys = None
        if pair is not None:
            ya_flat, ys = pair
            from pfrl_amd import ops

            if ops.dueling_softmax_supported(ya_flat, self.n_atoms):
                q = ops.dueling_softmax(ya_flat, ys, self.n_actions, self.n_atoms)
                return action_value.DistributionalDiscreteActionValue(q, self.z_values)
            h_a = h_v = None
        elif h.is_cuda and h.shape[1] % 2 == 0:
            # both halves contiguous out of ONE copy (the linear kernels want dense rows; chunk()
            # gives two strided views and each stream would copy its own: one launch fewer forwards
            # and one backwards, same values)
            halves = h.view(batch_size, 2, h.shape[1] // 2).transpose(0, 1).contiguous()
            h_a, h_v = halves[0], halves[1]
        else:
            h_a, h_v = torch.chunk(h, 2, dim=1)
        if pair is None:
            ya_flat = self.a_stream(h_a)
            if h.is_cuda:
                from pfrl_amd import ops

                if ops.dueling_softmax_supported(ya_flat, self.n_atoms):
                    # centring, value add and softmax over atoms in one launch
                    q = ops.dueling_softmax(ya_flat, self.v_stream(h_v), self.n_actions, self.n_atoms)
                    return action_value.DistributionalDiscreteActionValue(q, self.z_values)
        ya = ya_flat.reshape((batch_size, self.n_actions, self.n_atoms))
        mean = ya.sum(dim=1, keepdim=True) / self.n_actions
        ya = ya - mean
        ys = (ys if ys is not None else self.v_stream(h_v)).reshape((batch_size, 1, self.n_atoms))
        q = F.softmax(ya + ys, dim=2)
        return action_value.DistributionalDiscreteActionValue(q, self.z_values)
