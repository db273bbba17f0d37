"""Host-side observation wrappers for real (CPU) vector envs.

``LazyFrames`` / ``VectorFrameStack`` follow the reference
(/root/reference/pfrl/wrappers/atari_wrappers.py:251-272,
vector_frame_stack.py:54-105): consecutive observations share k-1 frame arrays
by identity and a reset fills the stack with k copies of the first frame.  The
device replay store relies on exactly that identity sharing to upload every
frame once (DeviceReplayStore._ingest_frame).  No gym dependency.
"""
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


def _forward_to_inner(name):
    def call(self, *args, **kwargs):
        return getattr(self.env, name)(*args, **kwargs)

    call.__name__ = name
    return call


class VectorEnvWrapper(VectorEnv):
    """A VectorEnv that forwards to the VectorEnv it wraps: the protocol methods one by one,
    every other public attribute through ``__getattr__`` (private names are never forwarded, so
    that a half-constructed wrapper fails with AttributeError instead of recursing)."""

    def __init__(self, env):
        self.env = env
        for attr in ("action_space", "observation_space"):
            setattr(self, attr, getattr(env, attr, None))

    def __getattr__(self, name):
        if name[:1] == "_":
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.env, name)

    unwrapped = property(lambda self: self.env.unwrapped)

    def __str__(self):
        return "<{}{}>".format(type(self).__name__, self.env)


for _name in ("step", "reset", "close", "render", "seed"):
    setattr(VectorEnvWrapper, _name, _forward_to_inner(_name))
del _name


class VectorFrameStack(VectorEnvWrapper):
    """The k last frames of every env as one observation (reference vector_frame_stack.py:54-105),
    kept the way the HBM frame ring keeps them (``pfrl_amd/device_store.py``): per env a RING of k
    frame references and a cursor.  A new frame overwrites the oldest slot; an observation is the k
    slots read in age order -- references, never copies, so consecutive observations share k - 1
    frame arrays by identity (what ``DeviceReplayStore`` de-duplicates on) and a reset writes the
    first frame into every slot."""

    def __init__(self, env, k, stack_axis=0):
        VectorEnvWrapper.__init__(self, env)
        self.k = k
        self.stack_axis = stack_axis
        n = env.num_envs
        self._slots = np.empty((n, k), dtype=object)     # frame references
        self._newest = np.full(n, -1, dtype=np.int64)    # slot written last, -1 = never reset
        self._age_order = (np.arange(k) + 1)             # + newest, mod k: oldest ... newest
        space = self.observation_space
        if hasattr(space, "low") and hasattr(space, "high"):
            # what one stacked observation looks like (reference vector_frame_stack.py:74-80)
            from pfrl_amd.spaces import Box

            self.observation_space = Box(np.repeat(space.low, k, axis=stack_axis),
                                         np.repeat(space.high, k, axis=stack_axis),
                                         dtype=space.dtype)

    def _push(self, e, frame):
        w = (self._newest[e] + 1) % self.k
        self._slots[e, w] = frame
        self._newest[e] = w

    def _observations(self):
        if (self._newest < 0).any():
            raise RuntimeError("VectorFrameStack: step() before every env was reset()")
        order = (self._newest[:, None] + self._age_order[None, :]) % self.k
        return [LazyFrames(list(self._slots[e, order[e]]), stack_axis=self.stack_axis)
                for e in range(len(order))]

    def reset(self, mask=None):
        firsts = self.env.reset(mask=mask)
        restarted = (np.ones(len(firsts), dtype=bool) if mask is None
                     else ~np.asarray(mask, dtype=bool))
        for e in np.flatnonzero(restarted):
            self._slots[e, :] = [firsts[e]] * self.k     # k references to the SAME frame
            self._newest[e] = self.k - 1
        return self._observations()

    def step(self, action):
        frames, reward, done, info = self.env.step(action)
        for e, frame in enumerate(frames):
            self._push(e, frame)
        return self._observations(), reward, done, info

    @property
    def frames(self):
        """Per env, the k frames oldest first (the reference's attribute of the same name)."""
        return [ob._frames for ob in self._observations()]


from pfrl_amd.wrappers.env_wrappers import (CastObservation, CastObservationToFloat32,  # NOQA,E402
                                            ContinuingTimeLimit, NormalizeActionSpace,
                                            RandomizeAction, Render, ScaleReward, Wrapper)
from pfrl_amd.wrappers import (atari_wrappers, cast_observation, continuing_time_limit,  # NOQA,E402
                               normalize_action_space, randomize_action, render, scale_reward,
                               vector_frame_stack)
