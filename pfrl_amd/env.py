"""Environment interfaces (reference pfrl/env.py:4-55)."""
from abc import ABCMeta, abstractmethod


class Env(object, metaclass=ABCMeta):
    @abstractmethod
    def step(self, action):
        raise NotImplementedError()

    @abstractmethod
    def reset(self):
        raise NotImplementedError()

    @abstractmethod
    def close(self):
        raise NotImplementedError()


class VectorEnv(object, metaclass=ABCMeta):
    """Batch of envs stepping in lockstep: ``step(actions)`` returns
    (observations, rewards, dones, infos); ``reset(mask)`` restarts the envs
    whose mask entry is False and returns all current observations."""

    @abstractmethod
    def step(self, actions):
        raise NotImplementedError()

    @abstractmethod
    def reset(self, mask):
        raise NotImplementedError()

    @abstractmethod
    def seed(self, seeds):
        raise NotImplementedError()

    @abstractmethod
    def close(self):
        raise NotImplementedError()

    @property
    def unwrapped(self):
        return self
