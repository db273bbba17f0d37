from pfrl_amd.explorers.epsilon_greedy import ConstantEpsilonGreedy, ExponentialDecayEpsilonGreedy, LinearDecayEpsilonGreedy  # NOQA
from pfrl_amd.explorers.greedy import Greedy  # NOQA
