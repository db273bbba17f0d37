"""Module path of the reference (pfrl/initializers/lecun_normal.py)."""
from pfrl_amd.initializers import init_lecun_normal  # NOQA
