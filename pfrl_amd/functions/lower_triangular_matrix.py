"""Batched lower-triangular matrices from their diagonal and strictly-lower entries
(reference pfrl/functions/lower_triangular_matrix.py): the Cholesky-style factor L(s) of the NAF
advantage matrix P(s) = L L^T.

``diag`` is (B, n); ``non_diag`` is (B, n(n-1)/2) with the entries below the diagonal in
row-major order ((1,0), (2,0), (2,1), (3,0), ...).  The result is built by scattering both into
one zero tensor through a flat index, so autograd sees two plain ``index_put`` writes.
"""
import torch


def _flat_positions(n, device):
    """Flat (row * n + col) positions of the diagonal and of the strictly-lower entries."""
    on = torch.arange(n, device=device) * (n + 1)
    rows, cols = torch.tril_indices(n, n, offset=-1, device=device)
    return on, rows * n + cols


def lower_triangular_matrix(diag, non_diag):
    assert isinstance(diag, torch.Tensor) and isinstance(non_diag, torch.Tensor)
    batch, n = diag.shape
    on, below = _flat_positions(n, diag.device)
    flat = torch.zeros((batch, n * n), dtype=torch.float32, device=diag.device)
    flat[:, below] = non_diag
    flat[:, on] = diag
    return flat.view(batch, n, n)


def set_batch_diagonal(array, diag_val):
    """In place: the diagonals of a (B, n, n) batch (kept for callers of the reference's helper)."""
    n = array.shape[-1]
    assert array.shape[-2] == n
    idx = torch.arange(n, device=array.device)
    array[:, idx, idx] = diag_val


def set_batch_non_diagonal(array, non_diag_val):
    """In place: the strictly-lower entries of a (B, n, n) batch, row-major."""
    n = array.shape[-1]
    assert array.shape[-2] == n
    rows, cols = torch.tril_indices(n, n, offset=-1, device=array.device)
    array[:, rows, cols] = non_diag_val
