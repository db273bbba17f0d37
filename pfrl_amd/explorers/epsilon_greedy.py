"""Epsilon-greedy explorers (reference pfrl/explorers/epsilon_greedy.py:8-134).

The random draws stay on the host, on the legacy global NumPy stream and in
the same order as the reference (one ``np.random.rand()`` per env, plus
whatever ``random_action_func`` draws when it fires): they are a handful of
scalars per step and they define what "identical seeds" means."""
from logging import getLogger

import numpy as np

from pfrl_amd import explorer


def select_action_epsilon_greedily(epsilon, random_action_func, greedy_action_func):
    if np.random.rand() < epsilon:
        return random_action_func(), False
    return greedy_action_func(), True


class _EpsilonGreedyBase(explorer.Explorer):
    uses_action_value = False

    def compute_epsilon(self, t):
        return self.epsilon

    def select_action(self, t, greedy_action_func, action_value=None):
        self.epsilon = self.compute_epsilon(t)
        a, greedy = select_action_epsilon_greedily(self.epsilon, self.random_action_func,
                                                   greedy_action_func)
        self.logger.debug("t:%s a:%s %s", t, a, "greedy" if greedy else "non-greedy")
        return a


class ConstantEpsilonGreedy(_EpsilonGreedyBase):
    def __init__(self, epsilon, random_action_func, logger=getLogger(__name__)):
        assert 0 <= epsilon <= 1
        self.epsilon = epsilon
        self.random_action_func = random_action_func
        self.logger = logger

    def __repr__(self):
        return "ConstantEpsilonGreedy(epsilon={})".format(self.epsilon)


class LinearDecayEpsilonGreedy(_EpsilonGreedyBase):
    def __init__(self, start_epsilon, end_epsilon, decay_steps, random_action_func,
                 logger=getLogger(__name__)):
        assert 0 <= start_epsilon <= 1
        assert 0 <= end_epsilon <= 1
        assert decay_steps >= 0
        self.start_epsilon = start_epsilon
        self.end_epsilon = end_epsilon
        self.decay_steps = decay_steps
        self.random_action_func = random_action_func
        self.logger = logger
        self.epsilon = start_epsilon

    def compute_epsilon(self, t):
        if t > self.decay_steps:
            return self.end_epsilon
        return self.start_epsilon + (self.end_epsilon - self.start_epsilon) * (t / self.decay_steps)

    def __repr__(self):
        return "LinearDecayEpsilonGreedy(epsilon={})".format(self.epsilon)


class ExponentialDecayEpsilonGreedy(_EpsilonGreedyBase):
    def __init__(self, start_epsilon, end_epsilon, decay, random_action_func,
                 logger=getLogger(__name__)):
        assert 0 <= start_epsilon <= 1
        assert 0 <= end_epsilon <= 1
        assert 0 < decay < 1
        self.start_epsilon = start_epsilon
        self.end_epsilon = end_epsilon
        self.decay = decay
        self.random_action_func = random_action_func
        self.logger = logger
        self.epsilon = start_epsilon

    def compute_epsilon(self, t):
        return max(self.start_epsilon * (self.decay ** t), self.end_epsilon)

    def __repr__(self):
        return "ExponentialDecayEpsilonGreedy(epsilon={})".format(self.epsilon)
