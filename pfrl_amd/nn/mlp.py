"""Multi-layer perceptron (reference pfrl/nn/mlp.py:7-36).

What has to match the reference for seeded parity is not only the function computed but the ORDER
in which the global torch RNG is consumed while the module is built: every ``nn.Linear`` draws
its default initialisation when it is constructed, and the LeCun-normal weights that actually
stay are drawn afterwards.  The order here is: construct all hidden layers, re-draw their weights
(zero biases), construct the output layer, re-draw its weight with ``last_wscale``.  With the same
seed this reproduces the reference's parameters bit for bit (the CartPole and SAC / TD3 / DDPG
traces under tests/golden start from such models).

Parameter names (``hidden_layers.<i>.weight|bias``, ``output.weight|bias``) are the reference's,
so its ``model.pt`` files load with ``strict=True``; with no hidden layer the ``ModuleList`` is
empty and contributes no keys.  ``hidden_layers`` / ``output`` are plain ``nn.Linear`` modules:
the NoisyNet conversion (``to_factorized_noisy``) swaps them in place by attribute name.
"""
import torch.nn as nn
import torch.nn.functional as F

from pfrl_amd.initializers import init_lecun_normal


class MLP(nn.Module):
    def __init__(self, in_size, out_size, hidden_sizes, nonlinearity=F.relu, last_wscale=1):
        super().__init__()
        self.in_size = in_size
        self.out_size = out_size
        self.hidden_sizes = hidden_sizes
        self.nonlinearity = nonlinearity
        sizes = [in_size] + list(hidden_sizes)
        self.hidden_layers = nn.ModuleList(
            [nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])])
        for layer in self.hidden_layers:
            init_lecun_normal(layer.weight)
            nn.init.zeros_(layer.bias)
        self.output = nn.Linear(sizes[-1], out_size)
        init_lecun_normal(self.output.weight, scale=last_wscale)
        nn.init.zeros_(self.output.bias)

    def forward(self, x):
        h = x
        for layer in self.hidden_layers:
            h = self.nonlinearity(layer(h))
        return self.output(h)
