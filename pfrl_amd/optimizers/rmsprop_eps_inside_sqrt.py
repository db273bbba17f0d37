"""Module path of the reference (pfrl/optimizers/rmsprop_eps_inside_sqrt.py)."""
from pfrl_amd.optimizers import RMSpropEpsInsideSqrt, SharedRMSpropEpsInsideSqrt  # NOQA
