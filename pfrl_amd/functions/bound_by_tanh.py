"""Module path of the reference (pfrl/functions/bound_by_tanh.py)."""
from pfrl_amd.nn.concat_obs_and_action import bound_by_tanh  # NOQA
