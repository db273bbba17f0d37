"""Abstract Q-function interfaces (reference pfrl/q_function.py).

``StateQFunction``: ``q(x) -> ActionValue`` over all actions of a state batch.
``StateActionQFunction``: ``q(x, a) -> Q-values`` of given state-action pairs."""
from abc import ABCMeta, abstractmethod


class StateQFunction(object, metaclass=ABCMeta):
    @abstractmethod
    def __call__(self, x):
        raise NotImplementedError()


class StateActionQFunction(object, metaclass=ABCMeta):
    @abstractmethod
    def __call__(self, x, a):
        raise NotImplementedError()
