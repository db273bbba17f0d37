"""Evaluation hooks (reference pfrl/experiments/evaluation_hooks.py:12-101).

An evaluation hook is called by the ``Evaluator`` after every evaluation phase with
``(env, agent, evaluator, step, eval_stats, agent_stats, env_stats)`` and announces through three
class attributes which training drivers may use it; the drivers refuse the others up front."""
from pfrl_amd.experiments.hooks import EvaluationHook  # NOQA

try:
    import optuna
except ImportError:      # optional dependency, exactly as in the reference
    optuna = None


class OptunaPrunerHook(EvaluationHook):
    """Reports the mean evaluation score to an Optuna trial and raises ``optuna.TrialPruned``
    when the trial's pruner says so.  Not usable with the asynchronous trainer, whose workers'
    exceptions do not propagate."""

    support_train_agent = True
    support_train_agent_batch = True
    support_train_agent_async = False

    def __init__(self, trial):
        if optuna is None:
            raise RuntimeError("OptunaPrunerHook requires optuna installed.")
        self.trial = trial

    def __call__(self, env, agent, evaluator, step, eval_stats, agent_stats, env_stats):
        self.trial.report(eval_stats["mean"], step)
        if self.trial.should_prune():
            raise optuna.TrialPruned()
