"""Module path of the reference (pfrl/explorers/greedy.py)."""
from pfrl_amd.explorers import Greedy  # NOQA
