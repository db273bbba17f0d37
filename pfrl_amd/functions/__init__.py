"""Small tensor functions (reference pfrl/functions)."""
from pfrl_amd.nn.concat_obs_and_action import bound_by_tanh  # NOQA
