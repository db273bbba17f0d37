"""Small tensor functions (reference pfrl/functions): as in the reference, the package exposes
its MODULES -- ``pfrl.functions.bound_by_tanh.bound_by_tanh(x, low, high)``,
``pfrl.functions.lower_triangular_matrix.lower_triangular_matrix(diag, non_diag)``."""
from pfrl_amd.functions import bound_by_tanh, lower_triangular_matrix  # NOQA
