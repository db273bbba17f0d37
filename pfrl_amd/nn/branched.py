"""Parallel branches over one input (reference pfrl/nn/branched.py)."""
import torch


class Branched(torch.nn.Module):
    def __init__(self, *modules):
        super().__init__()
        self.child_modules = torch.nn.ModuleList(modules)

    def forward(self, *args, **kwargs):
        return tuple(mod(*args, **kwargs) for mod in self.child_modules)
