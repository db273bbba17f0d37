"""Small tensor functions (reference pfrl/functions)."""
from pfrl_amd.nn.concat_obs_and_action import bound_by_tanh  # NOQA
from pfrl_amd.functions.lower_triangular_matrix import lower_triangular_matrix  # NOQA
