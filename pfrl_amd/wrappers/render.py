"""Module path of the reference (pfrl/wrappers/render.py); see env_wrappers."""
from pfrl_amd.wrappers.env_wrappers import Render  # NOQA
