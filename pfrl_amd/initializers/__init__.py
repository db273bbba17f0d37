"""Weight initialisers (reference pfrl/initializers: LeCun normal as the
chainer default, lecun_normal.py:5-10, chainer_default.py:9-21)."""
import numpy as np
import torch
import torch.nn as nn


def init_lecun_normal(tensor, scale=1.0):
    fan_in = torch.nn.init._calculate_correct_fan(tensor, "fan_in")
    std = scale * np.sqrt(1.0 / fan_in)
    with torch.no_grad():
        return tensor.normal_(0, std)


@torch.no_grad()
def init_chainer_default(layer):
    assert isinstance(layer, nn.Module)
    if isinstance(layer, (nn.Linear, nn.Conv2d)):
        init_lecun_normal(layer.weight)
        if layer.bias is not None:
            nn.init.zeros_(layer.bias)
    return layer
from pfrl_amd.initializers import chainer_default, lecun_normal  # NOQA,E402  (reference module paths)
