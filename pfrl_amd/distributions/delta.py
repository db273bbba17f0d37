"""Module path of the reference (pfrl/distributions/delta.py)."""
from pfrl_amd.distributions import Delta  # NOQA
