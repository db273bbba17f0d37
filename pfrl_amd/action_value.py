"""Q-function outputs (reference pfrl/action_value.py: ``ActionValue`` :8-41,
``DiscreteActionValue`` :44-94, ``DistributionalDiscreteActionValue`` :97-180).
These stay stock PyTorch: argmax / gather on the network output."""
from abc import ABCMeta, abstractmethod

import torch
import torch.nn.functional as F


class ActionValue(object, metaclass=ABCMeta):
    @property
    @abstractmethod
    def greedy_actions(self):
        raise NotImplementedError()

    @property
    @abstractmethod
    def max(self):
        raise NotImplementedError()

    @abstractmethod
    def evaluate_actions(self, actions):
        raise NotImplementedError()

    @property
    @abstractmethod
    def params(self):
        raise NotImplementedError()

    def __getitem__(self, i):
        raise NotImplementedError()


class DiscreteActionValue(ActionValue):
    """Q(s, .) for a discrete action space; q_values is (batch, n_actions)."""

    def __init__(self, q_values, q_values_formatter=lambda x: x):
        assert isinstance(q_values, torch.Tensor)
        self.device = q_values.device
        self.q_values = q_values
        self.n_actions = q_values.shape[1]
        self.q_values_formatter = q_values_formatter
        self._greedy = None
        self._max = None

    @property
    def greedy_actions(self):
        if self._greedy is None:
            self._greedy = self.q_values.detach().argmax(dim=1).int()
        return self._greedy

    @property
    def max(self):
        if self._max is None:
            index = self.greedy_actions.long().unsqueeze(1)
            self._max = self.q_values.gather(dim=1, index=index).flatten()
        return self._max

    def evaluate_actions(self, actions):
        return self.q_values.gather(dim=1, index=actions.long().unsqueeze(1)).flatten()

    def compute_advantage(self, actions):
        return self.evaluate_actions(actions) - self.max

    def compute_double_advantage(self, actions, argmax_actions):
        return self.evaluate_actions(actions) - self.evaluate_actions(argmax_actions)

    def compute_expectation(self, beta):
        return torch.sum(F.softmax(beta * self.q_values, dim=1) * self.q_values, dim=1)

    def __repr__(self):
        return "DiscreteActionValue greedy_actions:{} q_values:{}".format(
            self.greedy_actions.detach().cpu().numpy(),
            self.q_values_formatter(self.q_values.detach().cpu().numpy()))

    @property
    def params(self):
        return (self.q_values,)

    def __getitem__(self, i):
        return DiscreteActionValue(self.q_values[i], q_values_formatter=self.q_values_formatter)


class DistributionalDiscreteActionValue(ActionValue):
    """Categorical return distributions: q_dist (batch, n_actions, n_atoms),
    z_values (n_atoms,) (reference :97-180)."""

    def __init__(self, q_dist, z_values, q_values_formatter=lambda x: x):
        assert isinstance(q_dist, torch.Tensor)
        assert isinstance(z_values, torch.Tensor)
        assert q_dist.ndim == 3
        assert z_values.ndim == 1
        assert q_dist.shape[2] == int(z_values.shape[0])
        self.z_values = z_values
        self.q_values = torch.matmul(q_dist, self.z_values)
        self.q_dist = q_dist
        self.n_actions = q_dist.shape[1]
        self.q_values_formatter = q_values_formatter
        self.device = q_dist.device
        self._greedy = None
        self._max = None

    @property
    def greedy_actions(self):
        if self._greedy is None:
            self._greedy = self.q_values.argmax(dim=1).detach()
        return self._greedy

    @property
    def max(self):
        if self._max is None:
            self._max = torch.gather(self.q_values, 1, self.greedy_actions[:, None])[:, 0]
        return self._max

    @property
    def max_as_distribution(self):
        rows = torch.arange(self.q_values.shape[0], device=self.q_dist.device)
        return self.q_dist[rows, self.greedy_actions.detach()]

    def evaluate_actions(self, actions):
        return torch.gather(self.q_values, 1, actions[:, None])[:, 0]

    def evaluate_actions_as_distribution(self, actions):
        rows = torch.arange(self.q_values.shape[0], device=self.q_dist.device)
        return self.q_dist[rows, actions]

    def compute_advantage(self, actions):
        return self.evaluate_actions(actions) - self.max

    def compute_double_advantage(self, actions, argmax_actions):
        return self.evaluate_actions(actions) - self.evaluate_actions(argmax_actions)

    def compute_expectation(self, beta):
        return (F.softmax(beta * self.q_values, dim=1) * self.q_values).sum(dim=1)

    def __repr__(self):
        return "DistributionalDiscreteActionValue greedy_actions:{} q_values:{}".format(
            self.greedy_actions.detach().cpu().numpy(),
            self.q_values_formatter(self.q_values.detach().cpu().numpy()))

    @property
    def params(self):
        return (self.q_dist,)

    def __getitem__(self, i):
        return DistributionalDiscreteActionValue(self.q_dist[i], self.z_values,
                                                 q_values_formatter=self.q_values_formatter)


class QuantileDiscreteActionValue(DiscreteActionValue):
    """Return quantiles per action: ``quantiles`` is (batch, n_taus, n_actions); the
    action values are their mean over the taus (reference :183-229)."""

    def __init__(self, quantiles, q_values_formatter=lambda x: x):
        assert quantiles.ndim == 3
        self.quantiles = quantiles
        super().__init__(quantiles.mean(1), q_values_formatter)

    def evaluate_actions_as_quantiles(self, actions):
        rows = torch.arange(self.quantiles.shape[0], dtype=torch.long,
                            device=self.quantiles.device)
        return self.quantiles[rows, :, actions.long()]

    def __repr__(self):
        return "QuantileDiscreteActionValue greedy_actions:{} q_values:{}".format(
            self.greedy_actions.detach().cpu().numpy(),
            self.q_values_formatter(self.q_values.detach().cpu().numpy()))

    @property
    def params(self):
        return (self.quantiles,)

    def __getitem__(self, i):
        return QuantileDiscreteActionValue(self.quantiles[i], self.q_values_formatter)
