"""Environment interfaces the drivers and agents are written against.

The contract is the reference's (pfrl/env.py): a single ``Env`` has
step/reset/close; a ``VectorEnv`` steps ``num_envs`` environments in lockstep,
``step(actions)`` returning ``(observations, rewards, dones, infos)`` and
``reset(mask)`` restarting exactly the environments whose mask entry is False
while handing back the current observation of every environment.  Device
vector envs (pfrl_amd.envs.SyntheticAtariVectorEnv) return a
``DeviceObsBatch`` as the observation container; host ones return lists.
"""
import abc


class Env(abc.ABC):
    """One environment."""

    @abc.abstractmethod
    def reset(self):
        """Start an episode; returns the first observation."""

    @abc.abstractmethod
    def step(self, action):
        """-> (observation, reward, done, info)"""

    @abc.abstractmethod
    def close(self):
        """Release resources."""


class VectorEnv(abc.ABC):
    """``num_envs`` environments advancing together."""

    @abc.abstractmethod
    def reset(self, mask):
        """Restart envs with ``mask[i] == False``; returns all observations."""

    @abc.abstractmethod
    def step(self, actions):
        """-> (observations, rewards, dones, infos), one entry per env."""

    @abc.abstractmethod
    def seed(self, seeds):
        """Seed every env."""

    @abc.abstractmethod
    def close(self):
        """Release resources."""

    @property
    def unwrapped(self):
        return self
