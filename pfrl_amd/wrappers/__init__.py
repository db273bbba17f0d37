"""Host-side observation wrappers for real (CPU) vector envs.

``LazyFrames`` / ``VectorFrameStack`` follow the reference
(/root/reference/pfrl/wrappers/atari_wrappers.py:251-272,
vector_frame_stack.py:54-105): consecutive observations share k-1 frame arrays
by identity and a reset fills the stack with k copies of the first frame.  The
device replay store relies on exactly that identity sharing to upload every
frame once (DeviceReplayStore._ingest_frame).  No gym dependency.
"""
from collections import deque

import numpy as np

from pfrl_amd.env import VectorEnv


class LazyFrames(object):
    """Array-like that concatenates its frames only when converted."""

    def __init__(self, frames, stack_axis=2):
        self.stack_axis = stack_axis
        self._frames = frames

    def __array__(self, dtype=None, copy=None):
        out = np.concatenate(self._frames, axis=self.stack_axis)
        if dtype is not None:
            out = out.astype(dtype)
        return out


class VectorEnvWrapper(VectorEnv):
    """VectorEnv analog of gym.Wrapper."""

    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def close(self):
        return self.env.close()

    def render(self, *args, **kwargs):
        return self.env.render(*args, **kwargs)

    def seed(self, seed=None):
        return self.env.seed(seed)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __str__(self):
        return "<{}{}>".format(type(self).__name__, self.env)


class VectorFrameStack(VectorEnvWrapper):
    """Stack the k last frames of every env of a VectorEnv."""

    def __init__(self, env, k, stack_axis=0):
        VectorEnvWrapper.__init__(self, env)
        self.k = k
        self.stack_axis = stack_axis
        self.frames = [deque([], maxlen=k) for _ in range(env.num_envs)]
        space = self.observation_space
        if hasattr(space, "low") and hasattr(space, "high"):
            # what one stacked observation looks like (reference vector_frame_stack.py:74-80)
            from pfrl_amd.spaces import Box

            self.observation_space = Box(np.repeat(space.low, k, axis=stack_axis),
                                         np.repeat(space.high, k, axis=stack_axis),
                                         dtype=space.dtype)

    def reset(self, mask=None):
        batch_ob = self.env.reset(mask=mask)
        if mask is None:
            mask = np.zeros(self.env.num_envs)
        for m, frames, ob in zip(mask, self.frames, batch_ob):
            if not m:
                for _ in range(self.k):
                    frames.append(ob)
        return self._get_ob()

    def step(self, action):
        batch_ob, reward, done, info = self.env.step(action)
        for frames, ob in zip(self.frames, batch_ob):
            frames.append(ob)
        return self._get_ob(), reward, done, info

    def _get_ob(self):
        assert len(self.frames) == self.env.num_envs
        assert len(self.frames[0]) == self.k
        return [LazyFrames(list(frames), stack_axis=self.stack_axis) for frames in self.frames]


from pfrl_amd.wrappers.env_wrappers import (CastObservation, CastObservationToFloat32,  # NOQA,E402
                                            ContinuingTimeLimit, NormalizeActionSpace,
                                            RandomizeAction, Render, ScaleReward, Wrapper)
from pfrl_amd.wrappers import (atari_wrappers, cast_observation, continuing_time_limit,  # NOQA,E402
                               normalize_action_space, randomize_action, render, scale_reward,
                               vector_frame_stack)
