"""Single-env wrappers used by the training scripts, without gym.

The reference builds these on ``gym.Wrapper`` (pfrl/wrappers/continuing_time_limit.py,
cast_observation.py, scale_reward.py, randomize_action.py, normalize_action_space.py);
gym is not available here, so :class:`Wrapper` is a minimal stand-in with the same
delegation rules (unknown attributes are looked up on the wrapped env).  Any object
with ``reset() -> obs`` and ``step(a) -> (obs, reward, done, info)`` can be wrapped.
"""
import numpy as np

from pfrl_amd.spaces import Box


class Wrapper(object):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    def seed(self, seed=None):
        if hasattr(self.env, "seed"):
            return self.env.seed(seed)

    def close(self):
        if hasattr(self.env, "close"):
            return self.env.close()


class ContinuingTimeLimit(Wrapper):
    """Time limit that does NOT end the episode: past ``max_episode_steps`` the env keeps
    returning ``done=False`` and flags ``info['needs_reset'] = True``; the training loop
    resets it (and the agent treats the transition as non-terminal)."""

    def __init__(self, env, max_episode_steps):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["needs_reset"] = True
        return obs, reward, done, info


class CastObservation(Wrapper):
    """Observations cast to ``dtype`` (no copy when already of that type); the last raw
    observation is kept in ``original_observation``."""

    def __init__(self, env, dtype):
        super().__init__(env)
        self.dtype = dtype
        self.original_observation = None

    def observation(self, observation):
        self.original_observation = observation
        return observation.astype(self.dtype, copy=False)

    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self.observation(obs), reward, done, info


class CastObservationToFloat32(CastObservation):
    def __init__(self, env):
        super().__init__(env, np.float32)


class ScaleReward(Wrapper):
    """reward * scale; the unscaled value of the last step stays in ``original_reward``."""

    def __init__(self, env, scale):
        super().__init__(env)
        self.scale = scale
        self.original_reward = None

    def reward(self, reward):
        self.original_reward = reward
        return self.scale * reward

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs, self.reward(reward), done, info


class RandomizeAction(Wrapper):
    """With probability ``random_fraction`` the env receives a uniformly random discrete
    action instead of the agent's (evaluation-time stochasticity for Atari).  Uses its
    own RandomState, seeded through ``seed()``."""

    def __init__(self, env, random_fraction):
        super().__init__(env)
        assert 0 <= random_fraction <= 1
        assert hasattr(env.action_space, "n"), \
            "RandomizeAction supports only discrete action spaces"
        self._random_fraction = random_fraction
        self._np_random = np.random.RandomState()

    def action(self, action):
        if self._np_random.rand() < self._random_fraction:
            return self._np_random.randint(self.env.action_space.n)
        return action

    def step(self, action):
        return self.env.step(self.action(action))

    def seed(self, seed=None):
        super().seed(seed)
        self._np_random.seed(seed)


class NormalizeActionSpace(Wrapper):
    """The agent acts in [-1, 1]^n; actions are mapped affinely onto the env's box."""

    def __init__(self, env):
        super().__init__(env)
        space = env.action_space
        assert hasattr(space, "low") and hasattr(space, "high"), "needs a box action space"
        self.action_space = Box(-np.ones_like(space.low), np.ones_like(space.low),
                                dtype=np.asarray(space.low).dtype)

    def action(self, action):
        space = self.env.action_space
        return (np.asarray(action).copy() + 1) * ((space.high - space.low) / 2) + space.low

    def step(self, action):
        return self.env.step(self.action(action))


class Render(Wrapper):
    """Call ``env.render(**kwargs)`` after every reset and step (reference
    pfrl/wrappers/render.py:4-24)."""

    def __init__(self, env, **kwargs):
        super().__init__(env)
        self._kwargs = kwargs

    def reset(self, **kwargs):
        ret = self.env.reset(**kwargs)
        self.env.render(**self._kwargs)
        return ret

    def step(self, action):
        ret = self.env.step(action)
        self.env.render(**self._kwargs)
        return ret
