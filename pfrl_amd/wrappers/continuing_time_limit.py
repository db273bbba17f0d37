"""Module path of the reference (pfrl/wrappers/continuing_time_limit.py); see env_wrappers."""
from pfrl_amd.wrappers.env_wrappers import ContinuingTimeLimit  # NOQA
