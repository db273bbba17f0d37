from pfrl_amd.agents.dqn import DQN  # NOQA
from pfrl_amd.agents.double_dqn import DoubleDQN  # NOQA
from pfrl_amd.agents.ppo import PPO  # NOQA
from pfrl_amd.agents.a2c import A2C  # NOQA
from pfrl_amd.agents.categorical_dqn import CategoricalDQN, CategoricalDoubleDQN  # NOQA
from pfrl_amd.agents.soft_actor_critic import SoftActorCritic  # NOQA
from pfrl_amd.agents.td3 import TD3  # NOQA
from pfrl_amd.agents.ddpg import DDPG  # NOQA
from pfrl_amd.agents.iqn import IQN  # NOQA
from pfrl_amd.agents.advantage_learning import AL, PAL, DoublePAL  # NOQA
from pfrl_amd.agents.dpp import DPP, DPPL, DPPGreedy  # NOQA
from pfrl_amd.agents import al, categorical_double_dqn, double_pal, pal  # NOQA,E402  (reference module paths)
