"""Module path of the reference (pfrl/policies/softmax_policy.py)."""
from pfrl_amd.policies import SoftmaxCategoricalHead  # NOQA
