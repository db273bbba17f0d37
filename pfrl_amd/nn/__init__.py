from pfrl_amd.nn.atari_cnn import (LargeAtariCNN, SmallAtariCNN,  # NOQA
                                   fuse_conv_bias_relu)
from pfrl_amd.nn.mfma_trunk import accelerate_heads, fuse_sequential_trunk  # NOQA
from pfrl_amd.nn.mfma_linear import accelerate_mlp  # NOQA
from pfrl_amd.nn.branched import Branched  # NOQA
from pfrl_amd.nn.mlp import MLP  # NOQA
from pfrl_amd.nn.noisy_linear import FactorizedNoisyLinear, to_factorized_noisy  # NOQA
from pfrl_amd.nn.concat_obs_and_action import BoundByTanh, ConcatObsAndAction, Lambda  # NOQA
from pfrl_amd.nn import bound_by_tanh  # NOQA  (the MODULE, as in the reference)
from pfrl_amd.nn.empirical_normalization import EmpiricalNormalization  # NOQA
from pfrl_amd.nn.recurrent import Recurrent, RecurrentBranched, RecurrentSequential  # NOQA
from pfrl_amd.nn import lmbda, noisy_chain, recurrent_branched, recurrent_sequential  # NOQA,E402
