from pfrl_amd import explorer as _explorer
from pfrl_amd.explorers.additive_gaussian import AdditiveGaussian  # NOQA
from pfrl_amd.explorers.additive_ou import AdditiveOU  # NOQA
from pfrl_amd.explorers.boltzmann import Boltzmann  # NOQA
from pfrl_amd.explorers.epsilon_greedy import (ConstantEpsilonGreedy,  # NOQA
                                               ExponentialDecayEpsilonGreedy,
                                               LinearDecayEpsilonGreedy)


class Greedy(_explorer.Explorer):
    """Always takes the greedy action (no exploration; used by NoisyNet agents)."""

    uses_action_value = False

    def select_action(self, t, greedy_action_func, action_value=None):
        return greedy_action_func()

    def __repr__(self):
        return "Greedy()"
from pfrl_amd.explorers import greedy  # NOQA,E402  (reference module path)
