"""The A-B-C chain: the toy problem the reference's agent tests learn on
(reference pfrl/envs/abc.py:7-169), without gym.

``size`` states in a row; in state n only action n moves on, and taking all ``size`` correct
actions in a row pays +1.  A wrong action ends the episode (episodic) or leaves the state where
it is (continuing); after the last correct action the episode ends (episodic) or the chain
restarts at state 0 (continuing).  Observations are one-hot float32 vectors of length
``size + 2``: the states, the terminal state, and one spare slot -- with ``partially_observable``
some episodes show every observation shifted right by one, so that the agent has to remember how
the episode started.  With a continuous action space the action vector (clipped to [-1, 1]) is
read as logits over the ``size`` inner actions: arg-max if ``deterministic``, else one draw from
their softmax (``np.random.choice`` on the global stream, like the reference).
"""
import numpy as np

from pfrl_amd import env
from pfrl_amd.spaces import Box, Discrete


class ABC(env.Env):
    def __init__(self, size=2, discrete=True, partially_observable=False, episodic=True,
                 deterministic=False):
        self.size = size
        self.terminal_state = size
        self.episodic = episodic
        self.partially_observable = partially_observable
        self.deterministic = deterministic
        self.n_max_offset = 1
        self.n_dim_obs = size + 1 + self.n_max_offset
        self.observation_space = Box(-np.inf, np.inf, (self.n_dim_obs,))
        self.action_space = Discrete(size) if discrete else Box(-1.0, 1.0, (size,))
        self._continuous = not discrete

    def observe(self):
        onehot = np.zeros(self.n_dim_obs, dtype=np.float32)
        onehot[self._state + self._offset] = 1.0
        return onehot

    def reset(self):
        self._state = 0
        if not self.partially_observable:
            self._offset = 0
        elif self.deterministic:       # alternate: first episode shifted, second not, ...
            self._offset = (getattr(self, "_offset", 0) + 1) % (self.n_max_offset + 1)
        else:
            self._offset = np.random.randint(self.n_max_offset + 1)
        return self.observe()

    def _inner_action(self, action):
        assert isinstance(action, np.ndarray)
        logits = np.clip(action, self.action_space.low, self.action_space.high)
        if self.deterministic:
            return np.argmax(logits)
        weights = np.exp(logits)
        return np.random.choice(range(self.size), p=weights / weights.sum())

    def step(self, action):
        if self._continuous:
            action = self._inner_action(action)
        reward, done = 0, False
        if action != self._state:
            if self.episodic:
                done, self._state = True, self.terminal_state
        elif self._state < self.size - 1:
            self._state += 1
        else:
            reward = 1.0
            if self.episodic:
                done, self._state = True, self.terminal_state
            else:
                self._state = 0
        return self.observe(), reward, done, {}

    def close(self):
        pass
