"""Module path of the reference (pfrl/nn/noisy_chain.py)."""
from pfrl_amd.nn.noisy_linear import to_factorized_noisy  # NOQA
