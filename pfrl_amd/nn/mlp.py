"""Multi-layer perceptron (reference pfrl/nn/mlp.py)."""
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
