"""Test-only stand-in for the absent third-party ``gym`` package.

It exists solely so that the *reference* (pfnet/pfrl, mounted read-only at
/root/reference in the build container) can be imported by
``tests/golden/make_golden.py`` to record golden vectors.  It is NOT part of
the product and nothing under ``pfrl_amd/`` imports it.
"""
import types

import numpy as np

__version__ = "0.21.0"


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        super().__init__(tuple(shape), np.dtype(dtype))
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    __hash__ = None


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.dtype(np.int64))
        self.n = n

    def sample(self):
        return np.random.randint(self.n)

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n

    __hash__ = None


spaces = types.ModuleType("gym.spaces")
spaces.Space = Space
spaces.Box = Box
spaces.Discrete = Discrete


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        pass

    def close(self):
        pass

    def seed(self, seed=None):
        return [seed]

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.reward_range = getattr(env, "reward_range", None)
        self.metadata = getattr(env, "metadata", {})

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def render(self, mode="human", **kwargs):
        return self.env.render(mode, **kwargs)

    def close(self):
        return self.env.close()

    def seed(self, seed=None):
        return self.env.seed(seed)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i

    def observation(self, observation):
        raise NotImplementedError


class RewardWrapper(Wrapper):
    def step(self, action):
        o, r, d, i = self.env.step(action)
        return o, self.reward(r), d, i

    def reward(self, reward):
        raise NotImplementedError


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))

    def action(self, action):
        raise NotImplementedError


class _TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def step(self, action):
        o, r, d, i = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            i["TimeLimit.truncated"] = not d
            d = True
        return o, r, d, i

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


wrappers = types.ModuleType("gym.wrappers")
wrappers.TimeLimit = _TimeLimit


class _ScriptedAtari(Env):
    """An ALE-shaped stand-in for the reference's Atari example scripts (TEST SIDE ONLY; ALE is
    not installed): 210 x 160 RGB frames, lives, FIRE among the actions, rewards and game-overs
    from a private RandomState.  What examples/atari/*.py need from ``gym.make(
    '...NoFrameskip-v4')`` to run their whole pipeline (make_atari -> wrap_deepmind ->
    MultiprocessVectorEnv -> VectorFrameStack -> train_agent_batch_with_evaluation)."""

    def __init__(self, env_id):
        self.spec = types.SimpleNamespace(id=env_id, max_episode_steps=100000)
        self.action_space = Discrete(6)
        self.observation_space = Box(low=0, high=255, shape=(210, 160, 3), dtype=np.uint8)
        self._rs = np.random.RandomState(0)
        self.np_random = np.random.RandomState(1)
        self._lives = 0
        self._t = 0
        self.ale = types.SimpleNamespace(lives=lambda: self._lives)

    @property
    def unwrapped(self):
        return self

    def get_action_meanings(self):
        return ["NOOP", "FIRE", "RIGHT", "LEFT", "RIGHTFIRE", "LEFTFIRE"]

    def seed(self, seed=None):
        self._rs = np.random.RandomState(None if seed is None else int(seed) % (2 ** 32))
        self.np_random = np.random.RandomState(None if seed is None else (int(seed) + 1) % (2 ** 32))
        return [seed]

    def _frame(self):
        # a moving bright block on noise: cheap, and not constant under grey-scale / resize
        f = self._rs.randint(0, 64, size=(210, 160, 3)).astype(np.uint8)
        y, x = (7 * self._t) % 180, (11 * self._t) % 130
        f[y:y + 20, x:x + 20] = 255
        return f

    def reset(self):
        self._lives, self._t = 3, 0
        return self._frame()

    def step(self, action):
        self._t += 1
        u = self._rs.rand()
        reward = float(self._rs.choice([-1.0, 0.0, 0.0, 0.0, 2.0])) if int(action) % 2 else 0.0
        done = False
        if u < 0.002:
            done, self._lives = True, 0
        elif u < 0.01 and self._lives > 0:
            self._lives -= 1
            done = self._lives == 0
        return self._frame(), reward, done, {}


def make(env_id, *args, **kwargs):
    """Only CartPole exists here: pfrl_amd's numpy CartPole (gym's published dynamics) behind
    the terminating TimeLimit gym would put around it.  Enough for the reference's vector-env
    and wrapper tests to run under tools/run_reference_tests.py."""
    limits = {"CartPole-v0": 200, "CartPole-v1": 500}
    if "NoFrameskip" in env_id:
        return _TimeLimit(_ScriptedAtari(env_id), max_episode_steps=100000)
    if env_id not in limits:
        raise RuntimeError("gym shim: no environment %r" % (env_id,))
    from pfrl_amd.envs.cartpole import CartPoleEnv

    class _CartPole(Env):
        def __init__(self):
            self._env = CartPoleEnv(seed=0, max_episode_steps=10 ** 9)
            self.action_space = Discrete(2)
            high = np.asarray(self._env.observation_space.high, dtype=np.float32)
            self.observation_space = Box(-high, high, dtype=np.float32)
            self.spec = types.SimpleNamespace(id=env_id, max_episode_steps=limits[env_id])

        def seed(self, seed=None):
            self._env.seed(0 if seed is None else seed)
            return [seed]

        def reset(self):
            return self._env.reset()

        def step(self, action):
            return self._env.step(action)

    return _TimeLimit(_CartPole(), max_episode_steps=limits[env_id])


import sys  # noqa: E402

sys.modules.setdefault("gym.spaces", spaces)
sys.modules.setdefault("gym.wrappers", wrappers)
