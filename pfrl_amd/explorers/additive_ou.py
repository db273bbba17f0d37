"""Ornstein-Uhlenbeck action noise (https://arxiv.org/abs/1509.02971; reference
pfrl/explorers/additive_ou.py:8-66).  State x evolves as x += theta (mu - x) + N(0,
sigma); the first call draws x from the stationary distribution unless
``start_with_mu``.  Draws come from the global NumPy stream.

The noise state is shaped like the first action it sees and kept as float32; it is shared by all
calls, i.e. one process = one noise process (the reference's DDPG example explores a single env).
``evolve`` is public because tests and the reference's own callers step the process directly.
Order of draws per call: nothing on the first call with ``start_with_mu``; otherwise exactly one
``np.random.normal`` of the action's shape -- the stationary draw first, an Euler step after.
"""
from logging import getLogger

import numpy as np

from pfrl_amd import explorer


class AdditiveOU(explorer.Explorer):
    uses_action_value = False

    def __init__(self, mu=0.0, theta=0.15, sigma=0.3, start_with_mu=False,
                 logger=getLogger(__name__)):
        self.mu, self.theta, self.sigma = mu, theta, sigma
        self.start_with_mu = start_with_mu
        self.logger = logger
        self.ou_state = None

    def evolve(self):
        kick = np.random.normal(size=self.ou_state.shape, loc=0, scale=self.sigma)
        self.ou_state += self.theta * (self.mu - self.ou_state) + kick

    def select_action(self, t, greedy_action_func, action_value=None):
        a = greedy_action_func()
        if self.ou_state is not None:
            self.evolve()
        elif self.start_with_mu:
            self.ou_state = np.full(a.shape, self.mu, dtype=np.float32)
        else:
            stationary = self.sigma / np.sqrt(2 * self.theta - self.theta ** 2)
            self.ou_state = np.random.normal(size=a.shape, loc=self.mu,
                                             scale=stationary).astype(np.float32)
        self.logger.debug("t:%s noise:%s", t, self.ou_state)
        return a + self.ou_state

    def __repr__(self):
        return "AdditiveOU(mu={}, theta={}, sigma={})".format(self.mu, self.theta, self.sigma)
