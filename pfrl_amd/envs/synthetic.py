"""Synthetic Atari-shaped vector environments.

``SyntheticAtariVectorEnv`` keeps everything on the device: each step writes
one new (84, 84) uint8 frame per env straight into a ``DeviceFrameStore`` ring
with a counter-based generator kernel and returns a ``DeviceObsBatch`` -- k
frame slots per env with ``VectorFrameStack`` semantics (a reset fills the
stack with k copies of the first frame, pfrl/wrappers/vector_frame_stack.py:
83-91; consecutive observations share k-1 frames).  Rewards in {-1, 0, +1}
w.p. (0.05, 0.90, 0.05) and ``done`` w.p. 1/500 are pure functions of
(seed, env_id, t), evaluated on the host with the same hash, so the agent sees
host scalars exactly as it would from the reference's VectorEnv and no
device->host synchronisation is needed.

``HostSyntheticAtariVectorEnv`` is the CPU twin (numpy frames wrapped in a
LazyFrames-like object) used for the CPU baseline and for parity runs of the
reference agents.
"""
import numpy as np
import torch

from pfrl_amd import env as _env
from pfrl_amd import ops
from pfrl_amd.device_store import DeviceFrameStore, DeviceObsBatch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _u01(seed, env_ids, t, stream):
    """Counter-based uniform [0,1) per (seed, env, t, stream)."""
    with np.errstate(over="ignore"):
        key = _mix64(np.uint64(seed) ^ _mix64(env_ids.astype(np.uint64) * _G + np.uint64(t)))
        r = _mix64(key + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def reward_done_stream(seed, env_ids, t, p_done):
    u = _u01(seed, env_ids, t, 1)
    rewards = np.where(u < 0.05, -1.0, np.where(u < 0.10, 1.0, 0.0))
    dones = _u01(seed, env_ids, t, 2) < p_done
    return rewards, dones


class SyntheticAtariVectorEnv(_env.VectorEnv):
    def __init__(self, num_envs, device=None, store=None, frame_slots=None, seed=0, stack=4,
                 frame_shape=(84, 84), n_actions=6, p_done=1.0 / 500, env_id0=0):
        self.num_envs = int(num_envs)
        self.stack = int(stack)
        self.n_actions = n_actions
        self.p_done = p_done
        self.seed_value = int(seed)
        self.env_id0 = int(env_id0)
        if store is None:
            assert frame_slots is not None, "give a DeviceFrameStore or frame_slots"
            store = DeviceFrameStore(frame_slots, frame_shape, torch.uint8, device, stack=stack)
        self.store = store
        self.device = store.device
        self.env_ids = np.arange(self.env_id0, self.env_id0 + self.num_envs)
        self.t = 0
        self.refs = np.zeros((self.num_envs, self.stack), dtype=np.int32)
        self.seqs = np.zeros((self.num_envs, self.stack), dtype=np.int64)
        from pfrl_amd.staging import StagingRing

        self._stage = StagingRing(self.device, slot_bytes=max(1 << 14, 8 * self.num_envs * stack),
                                  n_slots=64)
        self._infos = [{} for _ in range(self.num_envs)]
        self.action_space_n = n_actions

    def seed(self, seeds):
        self.seed_value = int(seeds if np.isscalar(seeds) else seeds[0])

    def close(self):
        pass

    def _new_frames(self, idx):
        """Write one fresh frame for each env in ``idx``; returns (seqs, slots)."""
        n = len(idx)
        seqs, slots = self.store.alloc(n)
        if n == self.num_envs:
            # the whole batch takes consecutive ring positions: the kernel derives the slots
            ops.frames_synth_u8_ring(self.store.frames, int(seqs[0]), n, self.seed_value,
                                     self.env_id0, self.t)
        else:
            # subset (resets): one launch per contiguous run keeps env keys right
            (slots_dev,) = self._stage.upload([slots])
            start = 0
            while start < n:
                stop = start + 1
                while stop < n and idx[stop] == idx[stop - 1] + 1:
                    stop += 1
                ops.frames_synth_u8(self.store.frames, slots_dev[start:stop], self.seed_value,
                                    self.env_id0 + int(idx[start]), self.t)
                start = stop
        return seqs, slots

    def _obs(self):
        return DeviceObsBatch(self.store, self.refs.copy(), self.seqs.min(axis=1))

    def reset(self, mask=None):
        if mask is None:
            idx = np.arange(self.num_envs)
        else:
            idx = np.flatnonzero(~np.asarray(mask, dtype=bool))
        if len(idx):
            self.t += 1
            seqs, slots = self._new_frames(idx)
            self.refs[idx] = slots[:, None]
            self.seqs[idx] = seqs[:, None]
        return self._obs()

    def step(self, actions):
        self.t += 1
        seqs, slots = self._new_frames(np.arange(self.num_envs))
        self.refs[:, :-1] = self.refs[:, 1:]
        self.refs[:, -1] = slots
        self.seqs[:, :-1] = self.seqs[:, 1:]
        self.seqs[:, -1] = seqs
        rewards, dones = self._reward_done()
        return self._obs(), rewards, dones, self._infos

    def _reward_done(self):
        """``reward_done_stream`` for this env batch, evaluated natively on the host (same
        hash; tests/test_host_plan.py compares the two)."""
        from pfrl_amd import _native

        n = self.num_envs
        rewards = np.empty(n, dtype=np.float64)
        dones = np.empty(n, dtype=np.bool_)
        _native.check(_native.lib().pfrl_synth_reward_done(
            self.seed_value % (1 << 64), self.env_id0, n, self.t, float(self.p_done),
            rewards.ctypes.data, dones.ctypes.data), "synth_reward_done")
        return rewards, dones


class HostLazyFrames(object):
    """LazyFrames duck type (``_frames`` list, concatenation on axis 0,
    reference atari_wrappers.py:251-272) for host-side synthetic envs."""

    def __init__(self, frames):
        self._frames = frames

    def __array__(self, dtype=None, copy=None):
        out = np.concatenate(self._frames, axis=0)
        return out.astype(dtype) if dtype is not None else out

    @property
    def shape(self):
        return (len(self._frames),) + self._frames[0].shape[1:]


class HostSyntheticAtariVectorEnv(_env.VectorEnv):
    """Same streams as SyntheticAtariVectorEnv for rewards / dones; frames are
    host numpy arrays from per-env RandomState (SURVEY.md 8d CPU baseline)."""

    def __init__(self, num_envs, seed=0, stack=4, frame_shape=(84, 84), n_actions=6,
                 p_done=1.0 / 500, frame_pool=None):
        self.num_envs = num_envs
        self.stack = stack
        self.frame_shape = (1,) + tuple(frame_shape)
        # frame_pool=P: frames are fresh copies out of P pre-drawn random frames, so
        # that a benchmark measures ingest, not numpy's generator (env cost ~ 0)
        self._pool = None
        if frame_pool:
            self._pool = np.random.RandomState(seed).randint(
                0, 256, size=(int(frame_pool),) + self.frame_shape).astype(np.uint8)
        self.n_actions = n_actions
        self.p_done = p_done
        self.seed_value = seed
        self.env_ids = np.arange(num_envs)
        self.rs = [np.random.RandomState(seed * num_envs + i) for i in range(num_envs)]
        self.t = 0
        self.frames = [None] * num_envs
        self._infos = [{} for _ in range(num_envs)]

    def seed(self, seeds):
        pass

    def close(self):
        pass

    def _frame(self, i):
        if self._pool is not None:
            return self._pool[(i * 7919 + self.t * 104729) % len(self._pool)].copy()
        return self.rs[i].randint(0, 256, size=self.frame_shape).astype(np.uint8)

    def reset(self, mask=None):
        idx = range(self.num_envs) if mask is None else np.flatnonzero(~np.asarray(mask, bool))
        if len(idx):
            self.t += 1
        for i in idx:
            f = self._frame(i)
            self.frames[i] = [f] * self.stack
        return [HostLazyFrames(list(fr)) for fr in self.frames]

    def step(self, actions):
        self.t += 1
        for i in range(self.num_envs):
            self.frames[i] = self.frames[i][1:] + [self._frame(i)]
        rewards, dones = reward_done_stream(self.seed_value, self.env_ids, self.t, self.p_done)
        return [HostLazyFrames(list(fr)) for fr in self.frames], rewards, dones, self._infos


class HostSyntheticVectorObsEnv(_env.VectorEnv):
    """MuJoCo-shaped synthetic env: float32 observations
    ~ N(0, 1) of size obs_dim, reward ~ N(0, 1), done w.p. p_done; continuous
    actions are ignored.  Deterministic in (seed, env, t)."""

    def __init__(self, num_envs, obs_dim=376, act_dim=17, seed=0, p_done=1.0 / 1000):
        self.num_envs = num_envs
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.p_done = p_done
        self.rs = [np.random.RandomState(seed * num_envs + i) for i in range(num_envs)]
        self.obs = [None] * num_envs
        self._infos = [{} for _ in range(num_envs)]

    def seed(self, seeds):
        pass

    def close(self):
        pass

    def _draw(self, i):
        return self.rs[i].randn(self.obs_dim).astype(np.float32)

    def reset(self, mask=None):
        idx = range(self.num_envs) if mask is None else np.flatnonzero(~np.asarray(mask, bool))
        for i in idx:
            self.obs[i] = self._draw(i)
        return list(self.obs)

    def step(self, actions):
        rewards = np.zeros(self.num_envs)
        dones = np.zeros(self.num_envs, dtype=bool)
        for i in range(self.num_envs):
            self.obs[i] = self._draw(i)
            rewards[i] = self.rs[i].randn()
            dones[i] = self.rs[i].rand() < self.p_done
        return list(self.obs), rewards, dones, self._infos
