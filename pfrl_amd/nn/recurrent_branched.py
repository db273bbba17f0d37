"""Module path kept for ``from pfrl.nn.recurrent_branched import RecurrentBranched``."""
from pfrl_amd.nn.recurrent import RecurrentBranched  # NOQA
