"""Dynamic Policy Programming (https://arxiv.org/abs/1004.2027) on the DQN device path.

Mirrors ``pfrl.agents.dpp`` (/root/reference/pfrl/agents/dpp.py:9-136): the target is
``Q'(s, a) + r + discount (1 - terminal) L Q'(s') - L Q'(s)`` with an operator L that is
the Boltzmann expectation (DPP), the log-sum-exp (DPPL) or the max (DPPGreedy).
"""
from abc import ABCMeta, abstractmethod

import torch

from pfrl_amd.agents.dqn import DQN


class AbstractDPP(DQN, metaclass=ABCMeta):
    @abstractmethod
    def _l_operator(self, qout):
        raise NotImplementedError()

    def _compute_target_values(self, exp_batch):
        next_expect = self._l_operator(self._target_next_action_value(exp_batch))
        return exp_batch["reward"] + exp_batch["discount"] * (
            1 - exp_batch["is_state_terminal"]) * next_expect

    def _compute_y_and_t(self, exp_batch):
        n = exp_batch["reward"].shape[0]
        actions = exp_batch["action"]
        start = exp_batch.get("recurrent_state")
        batch_q = self._action_value(self.model, exp_batch["state"], start).evaluate_actions(
            actions).reshape((n, 1))
        with torch.no_grad():
            target_qout = self._action_value(self.target_model, exp_batch["state"], start)
            target_q = target_qout.evaluate_actions(actions).reshape((n, 1))
            here = self._l_operator(target_qout).reshape((n, 1))
            ahead = self._compute_target_values(exp_batch).reshape((n, 1))
            t = target_q + ahead - here
        return batch_q, t


class _EtaDPP(AbstractDPP):
    def __init__(self, *args, **kwargs):
        self.eta = kwargs.pop("eta", 1.0)
        super().__init__(*args, **kwargs)


class DPP(_EtaDPP):
    """L = Boltzmann-weighted expectation with inverse temperature ``eta``."""

    def _l_operator(self, qout):
        return qout.compute_expectation(self.eta)


class DPPL(_EtaDPP):
    """L = log-sum-exp(eta Q) / eta."""

    def _l_operator(self, qout):
        return torch.logsumexp(self.eta * qout.q_values, dim=1) / self.eta


class DPPGreedy(AbstractDPP):
    """L = max (the eta -> infinity limit)."""

    def _l_operator(self, qout):
        return qout.max
