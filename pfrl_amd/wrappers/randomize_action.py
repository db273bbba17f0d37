"""Module path of the reference (pfrl/wrappers/randomize_action.py); see env_wrappers."""
from pfrl_amd.wrappers.env_wrappers import RandomizeAction  # NOQA
