"""Step-hook and evaluation-hook interfaces (reference pfrl/experiments/hooks.py:6-23,
evaluation_hooks.py:12-52).  Any callable ``(env, agent, step)`` works as a step hook;
evaluation hooks declare which drivers they support through class attributes."""
from abc import ABCMeta, abstractmethod

from pfrl_amd.experiments.evaluator import LinearInterpolationHook  # NOQA  (re-export)


class StepHook(object, metaclass=ABCMeta):
    """Called by the training drivers after every (batched) env step."""

    @abstractmethod
    def __call__(self, env, agent, step):
        raise NotImplementedError


class EvaluationHook(object, metaclass=ABCMeta):
    """Called by :class:`Evaluator` after each evaluation phase with
    ``(env, agent, evaluator, step, eval_stats, agent_stats, env_stats)``."""

    support_train_agent = False
    support_train_agent_batch = False
    support_train_agent_async = False

    @abstractmethod
    def __call__(self, env, agent, evaluator, step, eval_stats, agent_stats, env_stats):
        raise NotImplementedError
