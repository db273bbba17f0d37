"""A scripted stand-in for an ALE game, shared by tests/golden/make_golden.py (which wraps it with
the REFERENCE's atari_wrappers) and tests/test_driver.py (which wraps it with pfrl_amd's).  Frames,
rewards, life losses and game-overs come from a private RandomState, so two instances with the
same seed produce the same game when they are driven with the same actions."""
import types

import numpy as np


class _Space:
    def __init__(self, shape, low, high, dtype):
        self.shape, self.dtype = shape, np.dtype(dtype)
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)


class FakeALE:
    def __init__(self, seed, frame_shape=(1, 6, 5), p_life=0.06, p_over=0.03):
        try:      # under the reference (make_golden.py) the wrappers insist on a gym Box
            from gym import spaces

            self.observation_space = spaces.Box(low=0, high=255, shape=frame_shape, dtype=np.uint8)
        except ImportError:
            self.observation_space = _Space(frame_shape, 0, 255, np.uint8)
        self.action_space = types.SimpleNamespace(n=3)
        self.np_random = np.random.RandomState(seed + 1000)   # what Noop / Flicker draw from
        self._rs = np.random.RandomState(seed)
        self._p_life, self._p_over = p_life, p_over
        self._lives = 0
        self.n_resets = self.n_steps = 0
        self.ale = types.SimpleNamespace(lives=lambda: self._lives)
        self.spec = None

    @property
    def unwrapped(self):
        return self

    def get_action_meanings(self):
        return ["NOOP", "FIRE", "UP"]

    def _frame(self):
        return self._rs.randint(0, 256, size=self.observation_space.shape).astype(np.uint8)

    def reset(self):
        self.n_resets += 1
        self._lives = 3
        return self._frame()

    def step(self, action):
        self.n_steps += 1
        r = self._rs.rand()
        reward = float(self._rs.choice([-3.0, 0.0, 0.0, 2.5])) + 0.1 * int(action)
        done = False
        if r < self._p_over:
            done = True
            self._lives = 0
        elif r < self._p_over + self._p_life and self._lives > 0:
            self._lives -= 1
            done = self._lives == 0
        return self._frame(), reward, done, {}
