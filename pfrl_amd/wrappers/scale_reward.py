"""Module path of the reference (pfrl/wrappers/scale_reward.py); see env_wrappers."""
from pfrl_amd.wrappers.env_wrappers import ScaleReward  # NOQA
