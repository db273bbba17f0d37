"""DeepMind-style Atari preprocessing for one env, without gym
(reference pfrl/wrappers/atari_wrappers.py:23-330, itself the OpenAI-baselines set).

This is the host side of SURVEY.md 8(f) row 3: what these wrappers emit -- ``LazyFrames`` over
``(1, 84, 84)`` uint8 frames shared by identity between consecutive observations -- is the format
the HBM frame ring ingests once per frame (``DeviceReplayStore.ingest``).  The wrappers are
duck-typed on ``reset() -> obs`` / ``step(a) -> (obs, reward, done, info)``; the ALE-specific ones
additionally read ``env.unwrapped.ale.lives()``, ``get_action_meanings()`` and
``np_random`` exactly where the reference does.  ``WarpFrame`` needs OpenCV and ``make_atari``
needs gym + ALE: both raise at the point of use when those are not installed.
"""
from collections import deque

import numpy as np

from pfrl_amd.spaces import Box
from pfrl_amd.wrappers import LazyFrames  # NOQA  (the reference exports it from here, :251-272)
from pfrl_amd.wrappers.env_wrappers import ContinuingTimeLimit, Wrapper

try:
    import cv2

    cv2.ocl.setUseOpenCL(False)
except Exception:    # not installed (or a broken native library)
    cv2 = None


def _over(done, info):
    return done or info.get("needs_reset", False)


class _ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self.observation(obs), reward, done, info


class _ScriptedReset(Wrapper):
    """Base of the wrappers that play a fixed action prefix after every reset.  If the game ends
    (or asks for a reset) inside the prefix it is reset again; ``_resume_from_reset`` says whether
    the observation of that second reset replaces the one of the interrupted step."""

    _resume_from_reset = True

    def _prefix(self):
        raise NotImplementedError

    def reset(self, **kwargs):
        obs = self.env.reset(**kwargs)
        for action in self._prefix():
            obs, _, done, info = self.env.step(action)
            if _over(done, info):
                fresh = self.env.reset(**kwargs)
                if self._resume_from_reset:
                    obs = fresh
        return obs

    def step(self, ac):
        return self.env.step(ac)


class NoopResetEnv(_ScriptedReset):
    """1..noop_max no-op steps (action 0) after every reset; the count is drawn from the env's
    own ``np_random`` unless ``override_num_noops`` is set (reference :23-54)."""

    def __init__(self, env, noop_max=30):
        super().__init__(env)
        assert env.unwrapped.get_action_meanings()[0] == "NOOP"
        self.noop_action = 0
        self.noop_max = noop_max
        self.override_num_noops = None

    def _prefix(self):
        noops = self.override_num_noops
        if noops is None:
            rng = self.unwrapped.np_random
            draw = rng.integers if hasattr(rng, "integers") else rng.randint   # Generator / RandomState
            noops = draw(1, self.noop_max + 1)
        assert noops > 0
        return [self.noop_action] * int(noops)


class FireResetEnv(_ScriptedReset):
    """Press FIRE (1) and then action 2 after a reset, for games that wait for it.  As in the
    reference (:57-75) the returned observation is the one of the last scripted step even when
    that step ended the game and triggered another reset."""

    _resume_from_reset = False

    def __init__(self, env):
        super().__init__(env)
        meanings = env.unwrapped.get_action_meanings()
        assert len(meanings) >= 3 and meanings[1] == "FIRE"

    def _prefix(self):
        return (1, 2)


class EpisodicLifeEnv(Wrapper):
    """A lost life ends the episode for the agent; the game itself is reset only when it is
    really over, otherwise ``reset`` advances it with one no-op (reference :78-115)."""

    def __init__(self, env):
        super().__init__(env)
        self.lives = 0
        self.needs_real_reset = True

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.needs_real_reset = _over(done, info)
        lives = self.env.unwrapped.ale.lives()
        # lives == 0 can persist for a few frames before the game reports done (Qbert)
        if 0 < lives < self.lives:
            done = True
        self.lives = lives
        return obs, reward, done, info

    def reset(self, **kwargs):
        if self.needs_real_reset:
            obs = self.env.reset(**kwargs)
        else:
            obs, _, _, _ = self.env.step(0)
        self.lives = self.env.unwrapped.ale.lives()
        return obs


class MaxAndSkipEnv(Wrapper):
    """Repeat the action ``skip`` times, sum the rewards and return the pixel-wise maximum of the
    last two frames of a full repeat (reference :118-146).  The two-frame buffer persists across
    steps, as in the reference: a repeat cut short by ``done`` maxes over older frames."""

    def __init__(self, env, skip=4):
        super().__init__(env)
        self._obs_buffer = np.zeros((2,) + tuple(env.observation_space.shape), dtype=np.uint8)
        self._skip = skip

    def step(self, action):
        total_reward = 0.0
        done = None
        for i in range(self._skip):
            obs, reward, done, info = self.env.step(action)
            tail = i - (self._skip - 2)
            if tail >= 0:
                self._obs_buffer[tail] = obs
            total_reward += reward
            if _over(done, info):
                break
        return self._obs_buffer.max(axis=0), total_reward, done, info


class ClipRewardEnv(Wrapper):
    """reward -> sign(reward) (reference :149-155)."""

    def reward(self, reward):
        return np.sign(reward)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs, self.reward(reward), done, info


class WarpFrame(_ObservationWrapper):
    """RGB -> 84x84 grayscale with OpenCV's area interpolation (reference :158-185)."""

    def __init__(self, env, channel_order="hwc"):
        if cv2 is None:
            raise RuntimeError("Cannot import cv2 module. Please install OpenCV-Python to use"
                               " WarpFrame.")
        super().__init__(env)
        self.width = self.height = 84
        shape = {"hwc": (84, 84, 1), "chw": (1, 84, 84)}[channel_order]
        self.observation_space = Box(0, 255, shape, np.uint8)

    def observation(self, frame):
        frame = cv2.cvtColor(frame, cv2.COLOR_RGB2GRAY)
        frame = cv2.resize(frame, (self.width, self.height), interpolation=cv2.INTER_AREA)
        return frame.reshape(self.observation_space.shape)


class FrameStack(Wrapper):
    """The last ``k`` frames as a ``LazyFrames``: consecutive observations share ``k - 1`` frame
    arrays by identity and a reset fills the stack with ``k`` references to the first frame
    (reference :188-222)."""

    def __init__(self, env, k, channel_order="hwc"):
        super().__init__(env)
        self.k = k
        self.frames = deque([], maxlen=k)
        self.stack_axis = {"hwc": 2, "chw": 0}[channel_order]
        space = env.observation_space
        self.observation_space = Box(np.repeat(space.low, k, axis=self.stack_axis),
                                     np.repeat(space.high, k, axis=self.stack_axis),
                                     dtype=space.dtype)

    def reset(self):
        ob = self.env.reset()
        self.frames.extend([ob] * self.k)
        return self._get_ob()

    def step(self, action):
        ob, reward, done, info = self.env.step(action)
        self.frames.append(ob)
        return self._get_ob(), reward, done, info

    def _get_ob(self):
        assert len(self.frames) == self.k
        return LazyFrames(list(self.frames), stack_axis=self.stack_axis)


class ScaledFloatFrame(_ObservationWrapper):
    """uint8 -> float32 / 255 on the host.  This materialises every observation (4x the bytes,
    no frame sharing); on the device path prefer the ``phi`` of the agent, which the gather
    kernels apply while expanding u8 -> f32 in HBM (reference :225-248)."""

    def __init__(self, env):
        super().__init__(env)
        self.scale = 255.0
        space = env.observation_space
        self.observation_space = Box(self.observation(space.low), self.observation(space.high),
                                     dtype=np.float32)

    def observation(self, observation):
        return np.array(observation).astype(np.float32) / self.scale


class FlickerFrame(_ObservationWrapper):
    """Blank the frame with probability 1/2, drawn from the env's ``np_random``
    (reference :275-285)."""

    def observation(self, observation):
        if self.unwrapped.np_random.rand() < 0.5:
            return np.zeros_like(observation)
        return observation


def make_atari(env_id, max_frames=30 * 60 * 60):
    """``gym.make`` + our own (continuing) time limit + no-op starts + frame skipping
    (reference :288-298)."""
    try:
        import gym
    except ImportError as e:
        raise ImportError("make_atari needs gym with the Atari environments installed") from e
    env = gym.make(env_id)
    assert "NoFrameskip" in env.spec.id
    assert isinstance(env, gym.wrappers.TimeLimit)
    env = env.env                  # drop gym's terminating TimeLimit
    if max_frames:
        env = ContinuingTimeLimit(env, max_episode_steps=max_frames)
    return MaxAndSkipEnv(NoopResetEnv(env, noop_max=30), skip=4)


def wrap_deepmind(env, episode_life=True, clip_rewards=True, frame_stack=True, scale=False,
                  fire_reset=False, channel_order="chw", flicker=False):
    """The Nature-DQN preprocessing stack, innermost first, in the reference's order (:301-330)."""
    wants_fire = fire_reset and "FIRE" in env.unwrapped.get_action_meanings()
    stack = [
        (episode_life, EpisodicLifeEnv, {}),
        (wants_fire, FireResetEnv, {}),
        (True, WarpFrame, dict(channel_order=channel_order)),
        (scale, ScaledFloatFrame, {}),
        (clip_rewards, ClipRewardEnv, {}),
        (flicker, FlickerFrame, {}),
        (frame_stack, FrameStack, dict(k=4, channel_order=channel_order)),
    ]
    for enabled, wrapper, kwargs in stack:
        if enabled:
            env = wrapper(env, **kwargs)
    return env
