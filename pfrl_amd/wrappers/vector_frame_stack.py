"""Module path of the reference (pfrl/wrappers/vector_frame_stack.py); the classes live in the package."""
from pfrl_amd.wrappers import VectorEnvWrapper, VectorFrameStack  # NOQA
