"""Module path of the reference (pfrl/agents/categorical_double_dqn.py)."""
from pfrl_amd.agents.categorical_dqn import CategoricalDoubleDQN  # NOQA
