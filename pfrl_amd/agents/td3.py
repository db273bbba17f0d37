"""Twin Delayed DDPG (http://arxiv.org/abs/1802.09477) on the device replay path.

Mirrors ``pfrl.agents.td3.TD3`` (/root/reference/pfrl/agents/td3.py): constructor
(:76-147), ``update_q_func`` (:168-219), ``update_policy`` (:221-237), ``update``
with the delayed policy / target update (:239-246), statistics (:319-328).  The
replay side is the HBM store shared with DQN / SAC (float32 vector observations
and actions); the whole update replays as a HIP graph, one graph for each of the
two step shapes (critics only; critics + policy + soft target sync).
"""
import copy
from logging import getLogger

import torch
from torch.nn import functional as F

from pfrl_amd.agents._replay_actor_critic import ReplayActorCritic
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_
from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.copy_param import soft_copy_params


def default_target_policy_smoothing_func(batch_action):
    """Clipped Gaussian noise on the target action, clipped to [-1, 1] (:20-23)."""
    noise = torch.clamp(0.2 * torch.randn_like(batch_action), -0.5, 0.5)
    return torch.clamp(batch_action + noise, -1, 1)


class TD3(ReplayActorCritic):
    saved_attributes = ("policy", "q_func1", "q_func2", "target_policy", "target_q_func1",
                        "target_q_func2", "policy_optimizer", "q_func1_optimizer",
                        "q_func2_optimizer")
    _STATS = (("q1", 1000), ("q2", 1000), ("loss1", 100), ("loss2", 100), ("policy_loss", 100))

    def __init__(self, policy, q_func1, q_func2, policy_optimizer, q_func1_optimizer,
                 q_func2_optimizer, replay_buffer, gamma, explorer, gpu=None,
                 replay_start_size=10000, minibatch_size=100, update_interval=1, phi=lambda x: x,
                 soft_update_tau=5e-3, n_times_update=1, max_grad_norm=None,
                 logger=getLogger(__name__), batch_states=batch_states, burnin_action_func=None,
                 policy_update_delay=2,
                 target_policy_smoothing_func=default_target_policy_smoothing_func,
                 use_graphs=None):
        self.policy, self.q_func1, self.q_func2 = policy, q_func1, q_func2
        self.policy_optimizer = policy_optimizer
        self.q_func1_optimizer = q_func1_optimizer
        self.q_func2_optimizer = q_func2_optimizer
        self.soft_update_tau = soft_update_tau
        self.max_grad_norm = max_grad_norm
        self.policy_update_delay = policy_update_delay
        self.target_policy_smoothing_func = target_policy_smoothing_func
        self.policy_n_updates = 0
        self.q_func_n_updates = 0
        # n_times_update is accepted and ignored, as in the reference (:118-126)
        self._setup([policy, q_func1, q_func2], gpu, replay_buffer, phi, gamma, explorer,
                    batch_states, logger, burnin_action_func, minibatch_size, replay_start_size,
                    update_interval, 1, use_graphs)
        frozen = lambda m: copy.deepcopy(m).eval().requires_grad_(False)
        self.target_policy = frozen(policy)
        self.target_q_func1 = frozen(q_func1)
        self.target_q_func2 = frozen(q_func2)
        from pfrl_amd.distributed import GradientAllReducer

        self._reducers = {m: GradientAllReducer(m) for m in (policy, q_func1, q_func2)}

    # -- hooks ---------------------------------------------------------------------
    def _policy(self):
        return self.policy

    def _burnin_over(self):
        return self.policy_n_updates > 0

    def _variant(self):
        return (self.q_func_n_updates + 1) % self.policy_update_delay == 0

    def _graph_modules(self):
        return [self.policy, self.q_func1, self.q_func2, self.target_policy, self.target_q_func1,
                self.target_q_func2]

    def _graph_optimizers(self):
        return [self.policy_optimizer, self.q_func1_optimizer, self.q_func2_optimizer]

    def sync_target_network(self):
        soft_copy_params([(self.target_policy, self.policy), (self.target_q_func1, self.q_func1),
                          (self.target_q_func2, self.q_func2)], self.soft_update_tau)

    # -- learning -----------------------------------------------------------------------
    def _step(self, loss, module, optimizer):
        optimizer.zero_grad()
        loss.backward()
        self._reducers[module].all_reduce()
        if self.max_grad_norm is not None:
            clip_l2_grad_norm_(module.parameters(), self.max_grad_norm)
        optimizer.step()

    def update_q_func(self, batch):
        next_state = batch["next_state"]
        with torch.no_grad(), evaluating(self.target_policy), evaluating(self.target_q_func1), \
                evaluating(self.target_q_func2):
            next_actions = self.target_policy_smoothing_func(
                self.target_policy(next_state).sample())
            next_q = torch.min(self.target_q_func1((next_state, next_actions)),
                               self.target_q_func2((next_state, next_actions)))
            target_q = batch["reward"] + batch["discount"] * (
                1.0 - batch["is_state_terminal"]) * torch.flatten(next_q)
        predict_q1 = torch.flatten(self.q_func1((batch["state"], batch["action"])))
        predict_q2 = torch.flatten(self.q_func2((batch["state"], batch["action"])))
        loss1 = F.mse_loss(target_q, predict_q1)
        loss2 = F.mse_loss(target_q, predict_q2)
        self._stat(q1=predict_q1, q2=predict_q2, loss1=loss1, loss2=loss2)
        self._step(loss1, self.q_func1, self.q_func1_optimizer)
        self._step(loss2, self.q_func2, self.q_func2_optimizer)

    def update_policy(self, batch):
        state = batch["state"]
        q = self.q_func1((state, self.policy(state).rsample()))
        loss = -torch.mean(q)
        self._stat(policy_loss=loss)
        self._step(loss, self.policy, self.policy_optimizer)

    def _update_impl(self, batch, with_policy):
        self.update_q_func(batch)
        if with_policy:
            self.update_policy(batch)
            self.sync_target_network()

    def _after_update(self, with_policy):
        self.q_func_n_updates += 1
        if with_policy:
            self.policy_n_updates += 1

    def get_statistics(self):
        return [
            ("average_q1", self._mean_stat("q1")),
            ("average_q2", self._mean_stat("q2")),
            ("average_q_func1_loss", self._mean_stat("loss1")),
            ("average_q_func2_loss", self._mean_stat("loss2")),
            ("average_policy_loss", self._mean_stat("policy_loss")),
            ("policy_n_updates", self.policy_n_updates),
            ("q_func_n_updates", self.q_func_n_updates),
        ]
