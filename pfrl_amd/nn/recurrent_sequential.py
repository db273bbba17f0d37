"""Module path kept for ``from pfrl.nn.recurrent_sequential import RecurrentSequential``."""
from pfrl_amd.nn.recurrent import RecurrentSequential  # NOQA
