"""Deep Deterministic Policy Gradients (https://arxiv.org/abs/1509.02971) on the
device replay path.

Mirrors ``pfrl.agents.ddpg.DDPG`` (/root/reference/pfrl/agents/ddpg.py):
constructor (:59-145), ``compute_critic_loss`` (:148-173: note the critic target
uses ``gamma`` itself, not the batch's n-step discount), ``compute_actor_loss``
(:175-188), ``update`` (:190-205), per-step target sync (:276-278), statistics
(:305-312).  With a Gaussian policy head this is SVG(0), as in the reference.
"""
import copy
from logging import getLogger

import torch
from torch import nn
from torch.nn import functional as F

from pfrl_amd.agents._replay_actor_critic import ReplayActorCritic
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.copy_param import synchronize_parameters


class DDPG(ReplayActorCritic):
    saved_attributes = ("model", "target_model", "actor_optimizer", "critic_optimizer")
    _STATS = (("q", 1000), ("actor_loss", 100), ("critic_loss", 100))

    def __init__(self, policy, q_func, actor_optimizer, critic_optimizer, replay_buffer, gamma,
                 explorer, gpu=None, replay_start_size=50000, minibatch_size=32,
                 update_interval=1, target_update_interval=10000, phi=lambda x: x,
                 target_update_method="hard", soft_update_tau=1e-2, n_times_update=1,
                 recurrent=False, episodic_update_len=None, logger=getLogger(__name__),
                 batch_states=batch_states, burnin_action_func=None, use_graphs=None):
        if recurrent:
            raise NotImplementedError("recurrent=True is not implemented (nor in the reference)")
        self.model = nn.ModuleList([policy, q_func])
        self.actor_optimizer = actor_optimizer
        self.critic_optimizer = critic_optimizer
        self.target_update_interval = target_update_interval
        self.target_update_method = target_update_method
        self.soft_update_tau = soft_update_tau
        self.recurrent = False
        self.n_updates = 0
        self._setup([self.model], gpu, replay_buffer, phi, gamma, explorer, batch_states, logger,
                    burnin_action_func, minibatch_size, replay_start_size, update_interval,
                    n_times_update, use_graphs)
        self.target_model = copy.deepcopy(self.model)
        self.target_model.eval()
        self.policy, self.q_function = self.model
        self.target_policy, self.target_q_function = self.target_model
        from pfrl_amd.distributed import GradientAllReducer

        self._reducers = {self.policy: GradientAllReducer(self.policy),
                          self.q_function: GradientAllReducer(self.q_function)}
        self.sync_target_network()

    # -- hooks ---------------------------------------------------------------------
    def _policy(self):
        return self.policy

    def _burnin_over(self):
        return self.n_updates > 0

    def _graph_modules(self):
        return [self.model, self.target_model]

    def _graph_optimizers(self):
        return [self.actor_optimizer, self.critic_optimizer]

    def _on_env_step(self):
        if self.t % self.target_update_interval == 0:
            self.sync_target_network()

    def sync_target_network(self):
        synchronize_parameters(src=self.model, dst=self.target_model,
                               method=self.target_update_method, tau=self.soft_update_tau)

    # -- learning -----------------------------------------------------------------------
    def compute_critic_loss(self, batch):
        n = batch["reward"].shape[0]
        with torch.no_grad():
            next_actions = self.target_policy(batch["next_state"]).sample()
            next_q = self.target_q_function((batch["next_state"], next_actions))
            target_q = batch["reward"] + self.gamma * (
                1.0 - batch["is_state_terminal"]) * next_q.reshape((n,))
        predict_q = self.q_function((batch["state"], batch["action"])).reshape((n,))
        loss = F.mse_loss(target_q, predict_q)
        self._stat(critic_loss=loss)
        return loss

    def compute_actor_loss(self, batch):
        state = batch["state"]
        q = self.q_function((state, self.policy(state).rsample()))
        loss = -q.mean()
        self._stat(q=q, actor_loss=loss)
        return loss

    def _update_impl(self, batch, variant=None):
        self.critic_optimizer.zero_grad()
        self.compute_critic_loss(batch).backward()
        self._reducers[self.q_function].all_reduce()
        self.critic_optimizer.step()
        self.actor_optimizer.zero_grad()
        self.compute_actor_loss(batch).backward()
        self._reducers[self.policy].all_reduce()
        self.actor_optimizer.step()

    def _after_update(self, variant=None):
        self.n_updates += 1

    def get_statistics(self):
        return [
            ("average_q", self._mean_stat("q")),
            ("average_actor_loss", self._mean_stat("actor_loss")),
            ("average_critic_loss", self._mean_stat("critic_loss")),
            ("n_updates", self.n_updates),
        ]
