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
        self.q_dist = q_dist
        self._q_values = None
        self.n_actions = q_dist.shape[1]
        self.q_values_formatter = q_values_formatter
        self.device = q_dist.device
        self._greedy = None
        self._max = None

    @property
    def q_values(self):
        """E[Z] per action (reference :113: computed in the constructor; here on first use -- the
        fused C51 loss reads the distributions only, and three matrix-vector launches per update
        were spent on expectations nobody looked at)."""
        if self._q_values is None:
            self._q_values = torch.matmul(self.q_dist, self.z_values)
        return self._q_values

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


class QuadraticActionValue(ActionValue):
    """Normalized advantage function (http://arxiv.org/abs/1603.00748; reference :231-325):
    ``Q(s, a) = V(s) - 1/2 (a - mu(s))^T P(s) (a - mu(s))`` with P positive definite, so the
    maximiser over a box is ``mu`` clipped to the box.

    ``mu`` (B, n), ``mat`` (B, n, n), ``v`` (B, 1); ``min_action`` / ``max_action`` are unbatched
    bounds (scalars or length-n sequences) or None."""

    def __init__(self, mu, mat, v, min_action=None, max_action=None):
        self.mu, self.mat, self.v = mu, mat, v
        self.device = mu.device
        self.batch_size = mu.shape[0]
        self.min_action = self._bound(min_action)
        self.max_action = self._bound(max_action)
        self._greedy = self._max = None

    def _bound(self, value):
        if value is None:
            return None
        if isinstance(value, (int, float)):
            value = [value]
        return torch.as_tensor(value).to(self.device).float()

    @property
    def greedy_actions(self):
        if self._greedy is None:
            a = self.mu
            if self.min_action is not None:
                a = torch.max(self.min_action.unsqueeze(0).expand_as(a), a)
            if self.max_action is not None:
                a = torch.min(self.max_action.unsqueeze(0).expand_as(a), a)
            self._greedy = a
        return self._greedy

    @property
    def max(self):
        if self._max is None:
            if self.min_action is None and self.max_action is None:
                self._max = self.v.reshape(self.batch_size)       # attained at mu itself
            else:
                self._max = self.evaluate_actions(self.greedy_actions)
        return self._max

    def evaluate_actions(self, actions):
        d = actions - self.mu
        quad = torch.matmul(torch.matmul(d[:, None, :], self.mat), d[:, :, None])[:, 0, 0]
        return self.v.reshape(self.batch_size) - 0.5 * quad

    def compute_advantage(self, actions):
        return self.evaluate_actions(actions) - self.max

    def compute_double_advantage(self, actions, argmax_actions):
        return self.evaluate_actions(actions) - self.evaluate_actions(argmax_actions)

    @property
    def params(self):
        return (self.mu, self.mat, self.v)

    def __getitem__(self, i):
        return QuadraticActionValue(self.mu[i], self.mat[i], self.v[i],
                                    min_action=self.min_action, max_action=self.max_action)

    def __repr__(self):
        return "QuadraticActionValue greedy_actions:{} v:{}".format(
            self.greedy_actions.detach().cpu().numpy(), self.v.detach().cpu().numpy())


class SingleActionValue(ActionValue):
    """Action value given as two callables: ``evaluator(actions) -> Q`` and, optionally,
    ``maximizer() -> greedy actions`` (reference :328-365).  Both are evaluated lazily, once."""

    def __init__(self, evaluator, maximizer=None):
        self.evaluator = evaluator
        self.maximizer = maximizer
        self._greedy = self._max = None

    @property
    def greedy_actions(self):
        if self._greedy is None:
            self._greedy = self.maximizer()
        return self._greedy

    @property
    def max(self):
        if self._max is None:
            self._max = self.evaluator(self.greedy_actions)
        return self._max

    def evaluate_actions(self, actions):
        return self.evaluator(actions)

    def compute_advantage(self, actions):
        return self.evaluator(actions) - self.max

    def compute_double_advantage(self, actions, argmax_actions):
        return self.evaluate_actions(actions) - self.evaluate_actions(argmax_actions)

    @property
    def params(self):
        import warnings

        warnings.warn("SingleActionValue has no learnable parameters until it is evaluated on "
                      "some action; use the tensor returned by evaluate_actions instead.")
        return ()

    def __getitem__(self, i):
        raise NotImplementedError

    def __repr__(self):
        return "SingleActionValue"
