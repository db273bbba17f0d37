"""Agent interfaces (mirrors /root/reference/pfrl/agent.py: ``Agent`` :9-70,
``AttributeSavingMixin`` :73-137, ``BatchAgent`` :157-200)."""
import contextlib
import os
from abc import ABCMeta, abstractmethod

import torch


class Agent(object, metaclass=ABCMeta):
    """Abstract agent."""

    training = True

    @abstractmethod
    def act(self, obs):
        raise NotImplementedError()

    @abstractmethod
    def observe(self, obs, reward, done, reset):
        raise NotImplementedError()

    @abstractmethod
    def save(self, dirname):
        pass

    @abstractmethod
    def load(self, dirname):
        pass

    @abstractmethod
    def get_statistics(self):
        pass

    @contextlib.contextmanager
    def eval_mode(self):
        orig_mode = self.training
        try:
            self.training = False
            yield
        finally:
            self.training = orig_mode


def _unwrap(module):
    if isinstance(module, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)):
        return module.module
    return module


class AttributeSavingMixin(object):
    """Save / load the attributes named in ``saved_attributes`` as
    ``<dirname>/<attr>.pt`` state dicts, recursing into nested mixins."""

    saved_attributes = ()

    def save(self, dirname):
        self._save_into(dirname, [])

    def _save_into(self, dirname, ancestors):
        os.makedirs(dirname, exist_ok=True)
        ancestors.append(self)
        for attr in self.saved_attributes:
            assert hasattr(self, attr)
            value = getattr(self, attr)
            if value is None:
                continue
            if isinstance(value, AttributeSavingMixin):
                assert not any(value is a for a in ancestors), "Avoid an infinite loop"
                value._save_into(os.path.join(dirname, attr), ancestors)
            else:
                torch.save(_unwrap(value).state_dict(), os.path.join(dirname, "%s.pt" % attr))
        ancestors.pop()

    def load(self, dirname):
        self._load_from(dirname, [])
        self._drop_captured_graphs()

    def _drop_captured_graphs(self):
        """Captured HIP graphs bake in the addresses of the optimizer's state tensors, and
        ``optimizer.load_state_dict`` replaces those tensors: after a load the graphs would
        keep stepping the old (freed) state.  Forget them; the next update captures again
        against the loaded state."""
        for name in ("_graphed", "_captured"):
            if getattr(self, name, None) is not None:
                setattr(self, name, None)

    def _load_from(self, dirname, ancestors):
        map_location = torch.device("cpu") if not torch.cuda.is_available() else None
        ancestors.append(self)
        for attr in self.saved_attributes:
            assert hasattr(self, attr)
            value = getattr(self, attr)
            if value is None:
                continue
            if isinstance(value, AttributeSavingMixin):
                assert not any(value is a for a in ancestors), "Avoid an infinite loop"
                value._load_from(os.path.join(dirname, attr), ancestors)
            else:
                _unwrap(value).load_state_dict(
                    torch.load(os.path.join(dirname, "%s.pt" % attr), map_location))
        ancestors.pop()


class BatchAgent(Agent, metaclass=ABCMeta):
    """Agent that can interact with a batch of envs; the single-env interface
    maps onto the batch one (reference :160-164)."""

    def act(self, obs):
        return self.batch_act([obs])[0]

    def observe(self, obs, reward, done, reset):
        self.batch_observe([obs], [reward], [done], [reset])

    @abstractmethod
    def batch_act(self, batch_obs):
        raise NotImplementedError()

    @abstractmethod
    def batch_observe(self, batch_obs, batch_reward, batch_done, batch_reset):
        raise NotImplementedError()
