"""Abstract policy interface (reference pfrl/policy.py): ``policy(state) -> action distribution``."""
from abc import ABCMeta, abstractmethod


class Policy(object, metaclass=ABCMeta):
    @abstractmethod
    def __call__(self, state):
        raise NotImplementedError()
