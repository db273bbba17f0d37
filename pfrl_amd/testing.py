"""Assertion helper for tests (reference pfrl/testing.py:5-23)."""
import numpy as np
import torch


def _to_numpy(x):
    """Tensors -> ndarrays, recursively through lists / tuples (which become arrays themselves)."""
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    if isinstance(x, (list, tuple)):
        return np.asarray([_to_numpy(item) for item in x])
    return x


def torch_assert_allclose(actual, desired, *args, **kwargs):
    """``numpy.testing.assert_allclose`` that accepts tensors (on any device) and nested
    lists / tuples of them on either side."""
    np.testing.assert_allclose(_to_numpy(actual), _to_numpy(desired), *args, **kwargs)
