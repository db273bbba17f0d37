"""Module path of the reference (pfrl/nn/lmbda.py)."""
from pfrl_amd.nn.concat_obs_and_action import Lambda  # NOQA
