"""Module path of the reference (pfrl/agents/pal.py)."""
from pfrl_amd.agents.advantage_learning import PAL  # NOQA
