from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv, SyntheticAtariVectorEnv  # NOQA
