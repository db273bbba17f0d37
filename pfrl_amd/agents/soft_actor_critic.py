"""Soft Actor-Critic on the device replay path (https://arxiv.org/abs/1812.05905).

Mirrors ``pfrl.agents.soft_actor_critic.SoftActorCritic``
(/root/reference/pfrl/agents/soft_actor_critic.py): constructor (:97-121),
``update_q_func`` (:214-262), ``update_policy_and_temperature`` (:273-308),
``update`` (:310-315), act / observe (:317-374), statistics (:376-385).
The replay side is the same HBM store as DQN with float32 vector observations
and float32 action vectors (one fused gather per minibatch, plain f32 copies);
statistics stay on the device instead of the reference's per-update
``.cpu().numpy()`` / ``.item()`` round trips (:241-244, :303-308).
"""
import copy
from logging import getLogger

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from pfrl_amd.agent import AttributeSavingMixin, BatchAgent
from pfrl_amd.agents.dqn import _DeviceRecord, _mean_or_nan
from pfrl_amd.replay_buffer import ReplayUpdater, batch_experiences
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_
from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.copy_param import synchronize_parameters
from pfrl_amd.utils.mode_of_distribution import mode_of_distribution


class TemperatureHolder(nn.Module):
    """Holds log(temperature) as a learnable scalar."""

    def __init__(self, initial_log_temperature=0):
        super().__init__()
        self.log_temperature = nn.Parameter(
            torch.tensor(initial_log_temperature, dtype=torch.float32))

    def forward(self):
        return torch.exp(self.log_temperature)


class SoftActorCritic(AttributeSavingMixin, BatchAgent):
    saved_attributes = ("policy", "q_func1", "q_func2", "target_q_func1", "target_q_func2",
                        "policy_optimizer", "q_func1_optimizer", "q_func2_optimizer",
                        "temperature_holder", "temperature_optimizer")

    def __init__(self, policy, q_func1, q_func2, policy_optimizer, q_func1_optimizer,
                 q_func2_optimizer, replay_buffer, gamma, gpu=None, replay_start_size=10000,
                 minibatch_size=100, update_interval=1, phi=lambda x: x, soft_update_tau=5e-3,
                 max_grad_norm=None, logger=getLogger(__name__), batch_states=batch_states,
                 burnin_action_func=None, initial_temperature=1.0, entropy_target=None,
                 temperature_optimizer_lr=None, act_deterministically=True, use_graphs=None):
        self.policy = policy
        self.q_func1 = q_func1
        self.q_func2 = q_func2
        if gpu is not None and gpu >= 0:
            assert torch.cuda.is_available()
            self.device = torch.device("cuda:{}".format(gpu))
            self.policy.to(self.device)
            self.q_func1.to(self.device)
            self.q_func2.to(self.device)
        else:
            self.device = torch.device("cpu")
        self.replay_buffer = replay_buffer
        if hasattr(replay_buffer, "bind"):
            replay_buffer.bind(self.device, phi)
        self.gamma = gamma
        self.gpu = gpu
        self.phi = phi
        self.soft_update_tau = soft_update_tau
        self.logger = logger
        self.policy_optimizer = policy_optimizer
        self.q_func1_optimizer = q_func1_optimizer
        self.q_func2_optimizer = q_func2_optimizer
        self.replay_updater = ReplayUpdater(
            replay_buffer=replay_buffer, update_func=self.update, batchsize=minibatch_size,
            n_times_update=1, replay_start_size=replay_start_size,
            update_interval=update_interval, episodic_update=False)
        self.max_grad_norm = max_grad_norm
        self.batch_states = batch_states
        self.burnin_action_func = burnin_action_func
        self.initial_temperature = initial_temperature
        self.entropy_target = entropy_target
        if self.entropy_target is not None:
            self.temperature_holder = TemperatureHolder(
                initial_log_temperature=np.log(initial_temperature))
            if temperature_optimizer_lr is not None:
                self.temperature_optimizer = torch.optim.Adam(
                    self.temperature_holder.parameters(), lr=temperature_optimizer_lr)
            else:
                self.temperature_optimizer = torch.optim.Adam(self.temperature_holder.parameters())
            self.temperature_holder.to(self.device)
        else:
            self.temperature_holder = None
            self.temperature_optimizer = None
        self.act_deterministically = act_deterministically
        self.t = 0
        self.target_q_func1 = copy.deepcopy(self.q_func1).eval().requires_grad_(False)
        self.target_q_func2 = copy.deepcopy(self.q_func2).eval().requires_grad_(False)
        self.q1_record = _DeviceRecord(1000)
        self.q2_record = _DeviceRecord(1000)
        self.entropy_record = _DeviceRecord(1000)
        self.q_func1_loss_record = _DeviceRecord(100)
        self.q_func2_loss_record = _DeviceRecord(100)
        self.n_policy_updates = 0
        from pfrl_amd import distributed
        from pfrl_amd.distributed import GradientAllReducer

        # HIP-graph capture of the whole update (two Q steps, policy step, temperature
        # step, soft target sync: ~250 small kernels, host-dispatch bound when eager).
        # Single GPU only: with world_size > 1 the three all-reduces stay eager.
        on_gpu = (self.device.type == "cuda" and getattr(replay_buffer, "is_device", False)
                  and distributed.world_size() == 1)
        self.use_graphs = on_gpu if use_graphs is None else bool(use_graphs and on_gpu)
        self._captured = None

        self._reducers = [GradientAllReducer(m) for m in (self.q_func1, self.q_func2, self.policy)]

    @property
    def temperature(self):
        if self.entropy_target is None:
            return self.initial_temperature
        with torch.no_grad():
            return float(self.temperature_holder())

    def _temperature_value(self):
        """Temperature as used inside the losses: a detached device scalar (no
        host round trip) or the fixed Python float."""
        if self.entropy_target is None:
            return self.initial_temperature
        with torch.no_grad():
            return self.temperature_holder().detach()

    def sync_target_network(self):
        synchronize_parameters(src=self.q_func1, dst=self.target_q_func1, method="soft",
                               tau=self.soft_update_tau)
        synchronize_parameters(src=self.q_func2, dst=self.target_q_func2, method="soft",
                               tau=self.soft_update_tau)

    def update_q_func(self, batch):
        batch_next_state = batch["next_state"]
        with torch.no_grad(), evaluating(self.policy), evaluating(self.target_q_func1), \
                evaluating(self.target_q_func2):
            next_action_distrib = self.policy(batch_next_state)
            next_actions = next_action_distrib.sample()
            next_log_prob = next_action_distrib.log_prob(next_actions)
            next_q1 = self.target_q_func1((batch_next_state, next_actions))
            next_q2 = self.target_q_func2((batch_next_state, next_actions))
            next_q = torch.min(next_q1, next_q2)
            entropy_term = self._temperature_value() * next_log_prob[..., None]
            assert next_q.shape == entropy_term.shape
            target_q = batch["reward"] + batch["discount"] * (
                1.0 - batch["is_state_terminal"]) * torch.flatten(next_q - entropy_term)
        predict_q1 = torch.flatten(self.q_func1((batch["state"], batch["action"])))
        predict_q2 = torch.flatten(self.q_func2((batch["state"], batch["action"])))
        loss1 = 0.5 * F.mse_loss(target_q, predict_q1)
        loss2 = 0.5 * F.mse_loss(target_q, predict_q2)
        self._stat(q1=predict_q1, q2=predict_q2, loss1=loss1, loss2=loss2)
        for loss, qf, opt, red in ((loss1, self.q_func1, self.q_func1_optimizer, self._reducers[0]),
                                   (loss2, self.q_func2, self.q_func2_optimizer, self._reducers[1])):
            opt.zero_grad()
            loss.backward()
            red.all_reduce()
            if self.max_grad_norm is not None:
                clip_l2_grad_norm_(qf.parameters(), self.max_grad_norm)
            opt.step()

    def update_temperature(self, log_prob):
        assert not log_prob.requires_grad
        loss = -torch.mean(self.temperature_holder() * (log_prob + self.entropy_target))
        self.temperature_optimizer.zero_grad()
        loss.backward()
        if self.max_grad_norm is not None:
            clip_l2_grad_norm_(self.temperature_holder.parameters(), self.max_grad_norm)
        self.temperature_optimizer.step()

    def update_policy_and_temperature(self, batch):
        batch_state = batch["state"]
        action_distrib = self.policy(batch_state)
        actions = action_distrib.rsample()
        log_prob = action_distrib.log_prob(actions)
        q1 = self.q_func1((batch_state, actions))
        q2 = self.q_func2((batch_state, actions))
        q = torch.min(q1, q2)
        entropy_term = self._temperature_value() * log_prob[..., None]
        assert q.shape == entropy_term.shape
        loss = torch.mean(entropy_term - q)
        self.policy_optimizer.zero_grad()
        loss.backward()
        self._reducers[2].all_reduce()
        if self.max_grad_norm is not None:
            clip_l2_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.policy_optimizer.step()
        if self._stat_sink is None:
            self.n_policy_updates += 1
        if self.entropy_target is not None:
            self.update_temperature(log_prob.detach())
        with torch.no_grad():
            try:
                ent = action_distrib.entropy()
            except NotImplementedError:
                ent = -log_prob
        self._stat(entropy=ent, policy_loss=loss)

    # -- statistics: recorded directly when eager, collected when capturing --------
    _stat_sink = None

    def _stat(self, **tensors):
        if self._stat_sink is not None:
            self._stat_sink.update({k: v.detach() for k, v in tensors.items()})
            return
        self._record_stats(tensors)

    def _record_stats(self, st):
        for name, rec in (("q1", self.q1_record), ("q2", self.q2_record),
                          ("loss1", self.q_func1_loss_record),
                          ("loss2", self.q_func2_loss_record), ("entropy", self.entropy_record)):
            if name in st:
                rec.extend(st[name])
        if "policy_loss" in st:
            self._last_policy_loss = st["policy_loss"].detach()

    _STAT_ORDER = ("q1", "q2", "loss1", "loss2", "entropy", "policy_loss")

    def _update_core(self, batch):
        """The captured step: returns all statistics as ONE flat device vector."""
        self._stat_sink = {}
        try:
            self.update_q_func(batch)
            self.update_policy_and_temperature(batch)
            self.sync_target_network()
            sink = self._stat_sink
        finally:
            self._stat_sink = None
        return {"stats": torch.cat([sink[k].reshape(-1).float() for k in self._STAT_ORDER]),
                "sizes": [sink[k].numel() for k in self._STAT_ORDER]}

    def _graph_capturable(self, batch):
        return (self.use_graphs and isinstance(batch.get("state"), torch.Tensor)
                and batch["state"].is_cuda)

    def update(self, experiences, errors_out=None):
        batch = batch_experiences(experiences, self.device, self.phi, self.gamma)
        if self._graph_capturable(batch):
            if self._captured is None:
                from pfrl_amd.agents.graphed_update import CapturedStep

                self._captured = CapturedStep(
                    self._update_core,
                    [self.policy, self.q_func1, self.q_func2, self.target_q_func1,
                     self.target_q_func2, self.temperature_holder],
                    [self.policy_optimizer, self.q_func1_optimizer, self.q_func2_optimizer,
                     self.temperature_optimizer], self.device)
            tensors = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
            try:
                out = self._captured.run(tensors)
            except Exception:
                self.logger.exception("HIP-graph capture of the SAC update failed; running eager")
                self.use_graphs = False
                self._captured = None
                return self.update(experiences, errors_out)
            self.n_policy_updates += 1
            flat = out["stats"].clone()   # the graph owns (and overwrites) its outputs
            pieces = torch.split(flat, out["sizes"])
            self._record_stats(dict(zip(self._STAT_ORDER, pieces)))
            return
        self.update_q_func(batch)
        self.update_policy_and_temperature(batch)
        self.sync_target_network()

    def batch_select_greedy_action(self, batch_obs, deterministic=False):
        with torch.no_grad(), evaluating(self.policy):
            batch_xs = self.batch_states(batch_obs, self.device, self.phi)
            policy_out = self.policy(batch_xs)
            if deterministic:
                return mode_of_distribution(policy_out).cpu().numpy()
            return policy_out.sample().cpu().numpy()

    def batch_act(self, batch_obs):
        if self.training:
            return self._batch_act_train(batch_obs)
        return self._batch_act_eval(batch_obs)

    def batch_observe(self, batch_obs, batch_reward, batch_done, batch_reset):
        if self.training:
            self._batch_observe_train(batch_obs, batch_reward, batch_done, batch_reset)

    def _batch_act_eval(self, batch_obs):
        assert not self.training
        return self.batch_select_greedy_action(batch_obs,
                                               deterministic=self.act_deterministically)

    def _batch_act_train(self, batch_obs):
        assert self.training
        if self.burnin_action_func is not None and self.n_policy_updates == 0:
            batch_action = [self.burnin_action_func() for _ in range(len(batch_obs))]
        else:
            batch_action = self.batch_select_greedy_action(batch_obs)
        self.batch_last_obs = list(batch_obs)
        self.batch_last_action = list(batch_action)
        return batch_action

    def _batch_observe_train(self, batch_obs, batch_reward, batch_done, batch_reset):
        assert self.training
        for i in range(len(batch_obs)):
            self.t += 1
            if self.batch_last_obs[i] is not None:
                assert self.batch_last_action[i] is not None
                self.replay_buffer.append(
                    state=self.batch_last_obs[i], action=self.batch_last_action[i],
                    reward=batch_reward[i], next_state=batch_obs[i], next_action=None,
                    is_state_terminal=batch_done[i], env_id=i)
                if batch_reset[i] or batch_done[i]:
                    self.batch_last_obs[i] = None
                    self.batch_last_action[i] = None
                    self.replay_buffer.stop_current_episode(env_id=i)
            self.replay_updater.update_if_necessary(self.t)

    def get_statistics(self):
        return [
            ("average_q1", _mean_or_nan(self.q1_record.values())),
            ("average_q2", _mean_or_nan(self.q2_record.values())),
            ("average_q_func1_loss", _mean_or_nan(self.q_func1_loss_record.values())),
            ("average_q_func2_loss", _mean_or_nan(self.q_func2_loss_record.values())),
            ("n_updates", self.n_policy_updates),
            ("average_entropy", _mean_or_nan(self.entropy_record.values())),
            ("temperature", self.temperature),
        ]
