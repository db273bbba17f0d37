"""Module path of the reference (pfrl/wrappers/normalize_action_space.py); see env_wrappers."""
from pfrl_amd.wrappers.env_wrappers import NormalizeActionSpace  # NOQA
