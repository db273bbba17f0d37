"""Explorer interface (reference pfrl/explorer.py)."""
from abc import ABCMeta, abstractmethod


class Explorer(object, metaclass=ABCMeta):
    @abstractmethod
    def select_action(self, t, greedy_action_func, action_value=None):
        raise NotImplementedError()
