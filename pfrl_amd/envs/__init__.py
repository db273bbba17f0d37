from pfrl_amd.envs.synthetic import (HostSyntheticAtariVectorEnv, HostSyntheticVectorObsEnv,  # NOQA
                                     SyntheticAtariVectorEnv)
from pfrl_amd.envs.multiprocess_vector_env import MultiprocessVectorEnv  # NOQA
from pfrl_amd.envs.serial_vector_env import SerialVectorEnv  # NOQA
from pfrl_amd.envs.cartpole import CartPoleEnv  # NOQA
from pfrl_amd.envs.abc import ABC  # NOQA
