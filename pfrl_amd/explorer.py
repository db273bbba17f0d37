"""Explorer interface (reference pfrl/explorer.py)."""
from abc import ABCMeta, abstractmethod


class Explorer(object, metaclass=ABCMeta):
    # Whether ``select_action`` reads its ``action_value`` argument.  Agents hand every env its
    # own one-row slice of the batched action value only to explorers that do (slicing a device
    # tensor per env is ~1 ms of host work per step at 256 envs); explorers written against the
    # reference without this attribute inherit True and get the reference's behaviour.
    uses_action_value = True

    @abstractmethod
    def select_action(self, t, greedy_action_func, action_value=None):
        raise NotImplementedError()
