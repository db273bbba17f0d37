"""Module path of the reference (pfrl/policies/deterministic_policy.py)."""
from pfrl_amd.policies import DeterministicHead  # NOQA
