"""Gradient clipping by global L2 norm (reference
pfrl/utils/clip_l2_grad_norm.py:5-38)."""
import torch


def clip_l2_grad_norm_(parameters, max_norm):
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    return torch.nn.utils.clip_grad_norm_(list(parameters), float(max_norm), norm_type=2)
