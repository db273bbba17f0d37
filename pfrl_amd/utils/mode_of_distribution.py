"""Most likely value of a torch distribution, used to act deterministically at evaluation time
(``act_deterministically=True`` in PPO / A2C / SAC; reference pfrl/utils/mode_of_distribution.py).

The result stays on the distribution's device.  Wrappers are peeled recursively:
``Independent`` does not change the mode, and a ``TransformedDistribution`` is handled by pushing
the base mode through its transforms -- exact for the monotone element-wise transforms the
policies here use (tanh squashing, affine), not a general statement about transformed densities.
"""
from torch import distributions as D


def _through_transforms(distrib):
    x = mode_of_distribution(distrib.base_dist)
    for transform in distrib.transforms:
        x = transform(x)
    return x


# first match wins; anything else (as in the reference) is an error rather than a guess
_MODE_RULES = (
    (D.Categorical, lambda d: d.probs.argmax(dim=-1)),
    ((D.Normal, D.MultivariateNormal), lambda d: d.mean),
    (D.Bernoulli, lambda d: (d.probs > 0.5).to(d.probs.dtype)),
    (D.Independent, lambda d: mode_of_distribution(d.base_dist)),
    (D.TransformedDistribution, _through_transforms),
)


def mode_of_distribution(distrib):
    for kinds, rule in _MODE_RULES:
        if isinstance(distrib, kinds):
            return rule(distrib)
    raise RuntimeError("{} is not supported".format(distrib))
