"""Batched lower-triangular matrices from their diagonal and strictly-lower entries
(reference pfrl/functions/lower_triangular_matrix.py): the Cholesky-style factor of the NAF
advantage matrix."""
import numpy as np
import torch


def set_batch_non_diagonal(array, non_diag_val):
    """Write (B, n(n-1)/2) values below the diagonal of every (n, n) matrix, row by row."""
    _, m, n = array.shape
    assert m == n
    rows, cols = np.tril_indices(n, -1)
    array[:, rows, cols] = non_diag_val


def set_batch_diagonal(array, diag_val):
    _, m, n = array.shape
    assert m == n
    idx = np.arange(n)
    array[:, idx, idx] = diag_val


def lower_triangular_matrix(diag, non_diag):
    assert isinstance(diag, torch.Tensor) and isinstance(non_diag, torch.Tensor)
    batch, n = diag.shape
    out = torch.zeros((batch, n, n), dtype=torch.float32, device=diag.device)
    set_batch_non_diagonal(out, non_diag)
    set_batch_diagonal(out, diag)
    return out
