"""Extra torch.distributions used by the deterministic-policy agents
(reference pfrl/distributions/delta.py)."""
from numbers import Number

import torch
from torch.distributions import Distribution, constraints


class Delta(Distribution):
    """Point mass at ``loc``: sampling returns ``loc`` (differentiably for
    ``rsample``); densities and entropy are undefined."""

    arg_constraints = {"loc": constraints.real}
    support = constraints.real
    has_rsample = True

    def __init__(self, loc, validate_args=None):
        self.loc = loc
        shape = torch.Size() if isinstance(loc, Number) else loc.size()
        super().__init__(shape, validate_args=validate_args)

    @property
    def mean(self):
        return self.loc

    @property
    def variance(self):
        return torch.zeros_like(self.loc)

    @property
    def stddev(self):
        return torch.zeros_like(self.loc)

    def expand(self, batch_shape, _instance=None):
        new = self._get_checked_instance(Delta, _instance)
        batch_shape = torch.Size(batch_shape)
        new.loc = self.loc.expand(batch_shape)
        super(Delta, new).__init__(batch_shape, validate_args=False)
        new._validate_args = self._validate_args
        return new

    def rsample(self, sample_shape=torch.Size()):
        return self.loc.expand(self._extended_shape(sample_shape))

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape).detach()

    def log_prob(self, value):
        raise RuntimeError("Not defined")

    def entropy(self):
        raise RuntimeError("Not defined")
from pfrl_amd.distributions import delta  # NOQA,E402  (reference module path)
