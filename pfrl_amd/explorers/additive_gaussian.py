"""Gaussian action noise (reference pfrl/explorers/additive_gaussian.py:6-36).
The noise is drawn from the global NumPy stream, one ``normal`` call per action,
exactly where the reference draws it."""
import numpy as np

from pfrl_amd import explorer


class AdditiveGaussian(explorer.Explorer):
    uses_action_value = False

    def __init__(self, scale, low=None, high=None):
        self.scale = scale
        self.low = low
        self.high = high

    def select_action(self, t, greedy_action_func, action_value=None):
        a = greedy_action_func()
        noisy = a + np.random.normal(scale=self.scale, size=a.shape).astype(np.float32)
        if self.low is None and self.high is None:
            return noisy
        return np.clip(noisy, self.low, self.high)

    def __repr__(self):
        return "AdditiveGaussian(scale={}, low={}, high={})".format(self.scale, self.low,
                                                                    self.high)
