"""pfrl_amd -- MI355X-native batched-RL hot path behind PFRL's Python API.

``import pfrl_amd as pfrl`` gives the names the reference's batched training
scripts use (agents, replay_buffers, experiments, explorers, q_functions, nn,
utils, wrappers are added as they are built; see SURVEY.md section 8 for the
scope contract and DESIGN.md for the data layout)."""
__version__ = "0.1.0"

from pfrl_amd import action_value  # NOQA
from pfrl_amd import agent  # NOQA
from pfrl_amd import agents  # NOQA
from pfrl_amd import distributions  # NOQA
from pfrl_amd import env  # NOQA
from pfrl_amd import envs  # NOQA
from pfrl_amd import experiments  # NOQA
from pfrl_amd import explorer  # NOQA
from pfrl_amd import explorers  # NOQA
from pfrl_amd import functions  # NOQA
from pfrl_amd import initializers  # NOQA
from pfrl_amd import nn  # NOQA
from pfrl_amd import optimizers  # NOQA
from pfrl_amd import policies  # NOQA
from pfrl_amd import policy  # NOQA
from pfrl_amd import q_function  # NOQA
from pfrl_amd import q_functions  # NOQA
from pfrl_amd import replay_buffer  # NOQA
from pfrl_amd import replay_buffers  # NOQA
from pfrl_amd import utils  # NOQA
from pfrl_amd import wrappers  # NOQA
