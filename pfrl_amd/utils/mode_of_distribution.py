"""Mode of a torch distribution (reference pfrl/utils/mode_of_distribution.py)."""
from torch import distributions as D


def mode_of_distribution(distrib):
    if isinstance(distrib, D.Categorical):
        return distrib.probs.argmax(dim=-1)
    if isinstance(distrib, (D.Normal, D.MultivariateNormal)):
        return distrib.mean
    if isinstance(distrib, D.Bernoulli):
        return (distrib.probs > 0.5).to(distrib.probs.dtype)
    if isinstance(distrib, D.Independent):
        return mode_of_distribution(distrib.base_dist)
    if isinstance(distrib, D.TransformedDistribution):
        x = mode_of_distribution(distrib.base_dist)
        for t in distrib.transforms:
            x = t(x)
        return x
    raise RuntimeError("{} is not supported".format(distrib))
