from pfrl_amd.envs.synthetic import (HostSyntheticAtariVectorEnv, HostSyntheticVectorObsEnv,  # NOQA
                                     SyntheticAtariVectorEnv)
