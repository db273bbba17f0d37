"""Boltzmann (softmax) exploration over discrete action values (reference
pfrl/explorers/boltzmann.py:8-30).

``select_action`` ignores the greedy action: it needs the action values themselves, so the agent
has to pass ``action_value`` (a ``DiscreteActionValue`` holding ONE row; the softmax is taken over
its flattened Q-values at temperature ``T``).  The action is one ``np.random.choice`` draw from
the global legacy NumPy stream -- the same stream epsilon-greedy and ``sample_n_k`` consume, which
is what makes seeded runs comparable with the reference.  The probabilities are computed on the
action value's device and cross to the host once per call; with hundreds of envs per step an
agent should prefer a batched sampler, this class exists for drop-in compatibility.
"""
import numpy as np
import torch

from pfrl_amd import action_value as _av
from pfrl_amd import explorer


class Boltzmann(explorer.Explorer):
    def __init__(self, T=1.0):
        self.T = T

    def select_action(self, t, greedy_action_func, action_value=None):
        assert action_value is not None
        assert isinstance(action_value, _av.DiscreteActionValue)
        with torch.no_grad():
            probs = torch.softmax(action_value.q_values / self.T, dim=-1).cpu().numpy().ravel()
        return np.random.choice(np.arange(action_value.q_values.shape[1]), p=probs)

    def __repr__(self):
        return "Boltzmann(T={})".format(self.T)
