"""CartPole-v1 without gym (BASELINE configs[0]; gym is not installed here).

The cart-pole of Barto, Sutton & Anderson (1983) with the constants gym
publishes for ``CartPole-v1``: gravity 9.8, cart 1.0 kg, pole 0.1 kg with half
length 0.5 m, +-10 N pushes, explicit Euler at 20 ms; an episode ends when |x| >
2.4 m or |theta| > 12 degrees; reward 1 per step; 500-step limit reported as
``info['needs_reset']`` (a truncation, not a terminal state).  State starts
uniform in [-0.05, 0.05]^4 from the env's own ``RandomState`` so that the
agent's global NumPy stream is untouched.
"""
import math

import numpy as np

from pfrl_amd.spaces import Box, Discrete


class CartPoleEnv:
    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, DT = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT = 2.4
    THETA_LIMIT = 12 * 2 * math.pi / 360

    def __init__(self, seed=0, max_episode_steps=500):
        self.rng = np.random.RandomState(seed)
        self.max_episode_steps = max_episode_steps
        self.action_space = Discrete(2, np.random.RandomState(seed + 1))
        hi = [2 * self.X_LIMIT, np.finfo(np.float32).max, 2 * self.THETA_LIMIT,
              np.finfo(np.float32).max]
        self.observation_space = Box([-v for v in hi], hi)
        self.state = None
        self._t = 0

    def seed(self, seed):
        self.rng = np.random.RandomState(seed)

    def reset(self):
        self.state = self.rng.uniform(-0.05, 0.05, size=4)
        self._t = 0
        return self.state.astype(np.float32)

    def step(self, action):
        assert action in (0, 1), action
        x, x_dot, th, th_dot = self.state
        f = self.FORCE if action == 1 else -self.FORCE
        total_m = self.M_CART + self.M_POLE
        pm_len = self.M_POLE * self.HALF_LEN
        c, s = math.cos(th), math.sin(th)
        tmp = (f + pm_len * th_dot * th_dot * s) / total_m
        th_acc = (self.GRAVITY * s - c * tmp) / (
            self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * c * c / total_m))
        x_acc = tmp - pm_len * th_acc * c / total_m
        x, x_dot = x + self.DT * x_dot, x_dot + self.DT * x_acc
        th, th_dot = th + self.DT * th_dot, th_dot + self.DT * th_acc
        self.state = np.array([x, x_dot, th, th_dot])
        self._t += 1
        done = bool(abs(x) > self.X_LIMIT or abs(th) > self.THETA_LIMIT)
        info = {}
        if not done and self.max_episode_steps and self._t >= self.max_episode_steps:
            info["needs_reset"] = True
        return self.state.astype(np.float32), 1.0, done, info

    def close(self):
        pass
