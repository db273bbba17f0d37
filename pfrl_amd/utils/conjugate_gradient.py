"""Conjugate gradient for ``A x = b`` with A given as a matrix-vector product
(reference pfrl/utils/conjugate_gradient.py; used by TRPO-style natural-gradient steps)."""
import torch


def conjugate_gradient(A_product_func, b, tol=1e-10, max_iter=10):
    """At most ``max_iter`` CG iterations from x = 0 for a symmetric positive-definite A; stops
    early once the residual norm drops below ``tol``.  Works on any device ``b`` lives on."""
    x = torch.zeros_like(b)
    residual = b - A_product_func(x)
    direction = residual
    rr = torch.matmul(residual, residual)
    for _ in range(max_iter):
        a_dir = A_product_func(direction)
        step = rr / torch.matmul(a_dir, direction)
        x = x + step * direction
        residual = residual - step * a_dir
        if torch.norm(residual) < tol:
            break
        rr_new = torch.matmul(residual, residual)
        direction = residual + (rr_new / rr) * direction
        rr = rr_new
    return x
