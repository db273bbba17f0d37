from pfrl_amd.nn.atari_cnn import LargeAtariCNN, SmallAtariCNN  # NOQA
from pfrl_amd.nn.branched import Branched  # NOQA
from pfrl_amd.nn.mlp import MLP  # NOQA
