"""Module path of the reference (pfrl/agents/double_pal.py)."""
from pfrl_amd.agents.advantage_learning import DoublePAL  # NOQA
