"""Boltzmann (softmax) exploration over discrete action values (reference
pfrl/explorers/boltzmann.py:8-30): one ``np.random.choice`` draw per action from the
global NumPy stream."""
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
