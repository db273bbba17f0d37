"""Module path of the reference (pfrl/initializers/chainer_default.py)."""
from pfrl_amd.initializers import init_chainer_default  # NOQA
