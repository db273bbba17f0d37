"""Advantage-learning variants of DQN (http://arxiv.org/abs/1512.04860): the Bellman
target is shifted by ``alpha`` times the action gap under the target network.

Mirrors ``pfrl.agents.al.AL`` (:7-79), ``pal.PAL`` (:7-80) and ``double_pal.DoublePAL``
(:7-71).  Only the target changes; replay, gathers and the captured update are DQN's
(these targets are not the plain TD target, so the composite torch loss is used).
"""
import torch

from pfrl_amd.agents import dqn
from pfrl_amd.utils.contexts import evaluating


class AL(dqn.DQN):
    """T_AL Q = T Q + alpha * (Q'(s, a) - max_b Q'(s, b))."""

    def __init__(self, *args, **kwargs):
        self.alpha = kwargs.pop("alpha", 0.9)
        super().__init__(*args, **kwargs)

    # pieces shared by the three operators ------------------------------------------------
    def _next_q(self, exp_batch, target_next_qout):
        return target_next_qout.max

    def _gap_term(self, cur_advantage, next_advantage):
        return cur_advantage

    def _compute_y_and_t(self, exp_batch):
        n = exp_batch["reward"].shape[0]
        actions = exp_batch["action"]
        start = exp_batch.get("recurrent_state")
        batch_q = self._action_value(self.model, exp_batch["state"], start).evaluate_actions(
            actions)
        with torch.no_grad():
            target_qout = self._action_value(self.target_model, exp_batch["state"], start)
            target_next_qout = self._target_next_action_value(exp_batch)
            next_q = self._next_q(exp_batch, target_next_qout).reshape(n)
            t_q = exp_batch["reward"] + exp_batch["discount"] * (
                1.0 - exp_batch["is_state_terminal"]) * next_q
            cur_adv = target_qout.compute_advantage(actions).reshape(n)
            next_adv = target_next_qout.compute_advantage(actions).reshape(n)
            target = t_q + self.alpha * self._gap_term(cur_adv, next_adv)
        return batch_q, target


class PAL(AL):
    """Persistent AL: the larger of the action gaps at s and at s'."""

    def _gap_term(self, cur_advantage, next_advantage):
        return torch.max(cur_advantage, next_advantage)


class DoublePAL(PAL):
    """PAL whose next value is Q'(s', argmax_b Q(s', b)) (Double-DQN selection)."""

    def _next_q(self, exp_batch, target_next_qout):
        with evaluating(self.model):
            next_qout = self._action_value(self.model, exp_batch["next_state"],
                                           exp_batch.get("next_recurrent_state"))
        return target_next_qout.evaluate_actions(next_qout.greedy_actions)
