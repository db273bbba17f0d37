"""Module path of the reference (pfrl/nn/bound_by_tanh.py): ``BoundByTanh`` and the function it applies."""
from pfrl_amd.nn.concat_obs_and_action import BoundByTanh, bound_by_tanh  # NOQA
