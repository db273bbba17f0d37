"""Soft Actor-Critic on the device replay path (https://arxiv.org/abs/1812.05905).

Mirrors ``pfrl.agents.soft_actor_critic.SoftActorCritic``
(/root/reference/pfrl/agents/soft_actor_critic.py): constructor (:97-121),
``update_q_func`` (:214-262), ``update_policy_and_temperature`` (:273-308),
``update`` (:310-315), act / observe (:317-374), statistics (:376-385).
The replay side is the same HBM store as DQN with float32 vector observations
and float32 action vectors (one fused gather per minibatch, plain f32 copies);
statistics stay on the device instead of the reference's per-update
``.cpu().numpy()`` / ``.item()`` round trips (:241-244, :303-308); the whole
update (two Q steps, policy step, temperature step, soft target sync) replays as
one HIP graph.  Acting / observing plumbing is shared with TD3 and DDPG
(:mod:`pfrl_amd.agents._replay_actor_critic`).
"""
import contextlib
import copy
import os
from logging import getLogger

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from pfrl_amd.agents import _sac_losses
from pfrl_amd.agents._replay_actor_critic import ReplayActorCritic
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_
from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.copy_param import soft_copy_params
from pfrl_amd.utils.mode_of_distribution import mode_of_distribution
from pfrl_amd.utils.squashed_gaussian import sample_with_log_prob


@contextlib.contextmanager
def _frozen(*modules):
    """requires_grad off for the parameters of ``modules`` while a graph is recorded."""
    params = [p for m in modules for p in m.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in params:
            p.requires_grad_(True)


class TemperatureHolder(nn.Module):
    """Holds log(temperature) as a learnable scalar."""

    def __init__(self, initial_log_temperature=0):
        super().__init__()
        self.log_temperature = nn.Parameter(
            torch.tensor(initial_log_temperature, dtype=torch.float32))

    def forward(self):
        return torch.exp(self.log_temperature)


class SoftActorCritic(ReplayActorCritic):
    saved_attributes = ("policy", "q_func1", "q_func2", "target_q_func1", "target_q_func2",
                        "policy_optimizer", "q_func1_optimizer", "q_func2_optimizer",
                        "temperature_holder", "temperature_optimizer")
    _STATS = (("q1", 1000), ("q2", 1000), ("entropy", 1000), ("loss1", 100), ("loss2", 100),
              ("policy_loss", 1))

    def __init__(self, policy, q_func1, q_func2, policy_optimizer, q_func1_optimizer,
                 q_func2_optimizer, replay_buffer, gamma, gpu=None, replay_start_size=10000,
                 minibatch_size=100, update_interval=1, phi=lambda x: x, soft_update_tau=5e-3,
                 max_grad_norm=None, logger=getLogger(__name__), batch_states=batch_states,
                 burnin_action_func=None, initial_temperature=1.0, entropy_target=None,
                 temperature_optimizer_lr=None, act_deterministically=True, use_graphs=None):
        self.policy, self.q_func1, self.q_func2 = policy, q_func1, q_func2
        self.policy_optimizer = policy_optimizer
        self.q_func1_optimizer = q_func1_optimizer
        self.q_func2_optimizer = q_func2_optimizer
        self.soft_update_tau = soft_update_tau
        self.max_grad_norm = max_grad_norm
        self.initial_temperature = initial_temperature
        self.entropy_target = entropy_target
        self.act_deterministically = act_deterministically
        self.n_policy_updates = 0
        self._setup([policy, q_func1, q_func2], gpu, replay_buffer, phi, gamma, None,
                    batch_states, logger, burnin_action_func, minibatch_size, replay_start_size,
                    update_interval, 1, use_graphs)
        if entropy_target is not None:
            self.temperature_holder = TemperatureHolder(
                initial_log_temperature=np.log(initial_temperature))
            kw = {} if temperature_optimizer_lr is None else {"lr": temperature_optimizer_lr}
            if self.device.type == "cuda":
                # torch's foreach Adam is 17 launches for this one scalar; same arithmetic
                from pfrl_amd.optimizers import FusedAdam as _Adam
            else:
                _Adam = torch.optim.Adam
            self.temperature_optimizer = _Adam(self.temperature_holder.parameters(), **kw)
            self.temperature_holder.to(self.device)
        else:
            self.temperature_holder = None
            self.temperature_optimizer = None
        frozen = lambda m: copy.deepcopy(m).eval().requires_grad_(False)
        self.target_q_func1 = frozen(q_func1)
        self.target_q_func2 = frozen(q_func2)
        from pfrl_amd.distributed import GradientAllReducer

        self._reducers = {m: GradientAllReducer(m) for m in (policy, q_func1, q_func2)}
        if self.temperature_holder is not None:
            # one more (scalar) all-reduce so that the replicas' temperatures stay equal
            self._reducers[self.temperature_holder] = GradientAllReducer(self.temperature_holder)
        self._policy_head = self._recognise_policy_head()

    def _recognise_policy_head(self):
        """(body, HeadSpec) when the policy is ``nn.Sequential(..., nn.Linear(_, 2A),
        Lambda(head))`` and ``head`` is recognised as the example's squashed-Gaussian head
        (utils/squashed_gaussian.recognise_head: probed on the device, never assumed): the
        update then runs ``body`` and one fused launch each way instead of the head's five
        elementwise launches forward and ten backward.  None: the policy is called as it is."""
        import os

        from pfrl_amd.nn import Lambda
        from pfrl_amd.utils.squashed_gaussian import recognise_head

        if self.device.type != "cuda" or os.environ.get("PFRL_SAC_FUSED_HEAD", "1") == "0":
            return None
        pol = self.policy
        if not isinstance(pol, nn.Sequential) or len(pol) < 2 or type(pol[-1]) is not Lambda:
            return None
        last = pol[-2]
        if not isinstance(last, nn.Linear):
            return None
        spec = recognise_head(pol[-1].lambd, last.out_features, self.device)
        if spec is None:
            return None
        body = nn.Sequential(*list(pol.children())[:-1])      # the same child modules, one fewer
        body.__class__ = type(pol) if isinstance(pol, nn.Sequential) else nn.Sequential
        self._policy_body = body        # (not a saved attribute: the same children as self.policy)
        return spec

    def _sample_policy(self, obs, reparameterize):
        """(actions, log_prob, -log_prob or None, distribution or None) of the policy on ``obs``."""
        if self._policy_head is not None:
            from pfrl_amd.utils.squashed_gaussian import head_sample_with_log_prob

            x = self._policy_body(obs)
            if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
                a, lp, neg = head_sample_with_log_prob(x, self._policy_head, reparameterize)
                return a, lp, neg, None
            distrib = self.policy[-1](x)
        else:
            distrib = self.policy(obs)
        a, lp, neg = sample_with_log_prob(distrib, reparameterize, with_negation=True)
        return a, lp, neg, distrib

    # reference attribute names of the statistics windows
    q1_record = property(lambda self: self._records["q1"])
    q2_record = property(lambda self: self._records["q2"])
    entropy_record = property(lambda self: self._records["entropy"])
    q_func1_loss_record = property(lambda self: self._records["loss1"])
    q_func2_loss_record = property(lambda self: self._records["loss2"])

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

    def _loss_temperature(self):
        """What the loss functions take: the fixed float, or the log-temperature parameter
        (they use exp of it, detached, as ``_temperature_value`` does)."""
        if self.entropy_target is None:
            return self.initial_temperature
        return self.temperature_holder.log_temperature

    # -- hooks -------------------------------------------------------------------------
    def _policy(self):
        return self.policy

    def _burnin_over(self):
        return self.n_policy_updates > 0

    def _graph_modules(self):
        return [self.policy, self.q_func1, self.q_func2, self.target_q_func1,
                self.target_q_func2, self.temperature_holder]

    def _graph_optimizers(self):
        return [self.policy_optimizer, self.q_func1_optimizer, self.q_func2_optimizer,
                self.temperature_optimizer]

    def sync_target_network(self):
        # (both target networks in one launch on the GPU)
        soft_copy_params([(self.target_q_func1, self.q_func1), (self.target_q_func2, self.q_func2)],
                         self.soft_update_tau)

    # -- learning ----------------------------------------------------------------------------
    def _defer_slabs(self, pairs):
        """May the backward pass of these (module, optimizer) pairs leave split-K gradient slabs
        unfolded for the optimizer launch to sum (nn.mfma_linear.slab_sink)?  Only when nothing
        reads ``.grad`` in between: no gradient all-reduce, no clipping, and the optimizer is
        exactly FusedAdam (whose step takes the slabs)."""
        from pfrl_amd.optimizers import FusedAdam

        if (self.device.type != "cuda" or self.max_grad_norm is not None
                or os.environ.get("PFRL_SAC_RIDERS", "1") == "0"):
            return False
        for module, optimizer in pairs:
            red = self._reducers.get(module)
            if type(optimizer) is not FusedAdam or (red is not None and red.active()):
                return False
        return True

    def _backward(self, losses, defer):
        """backward of scalar losses with dL/dL = 1 from a tensor kept around (backward() would
        fill a new one per loss; the loss nodes of _sac_losses recognise this one and skip their
        own backward launch).  Returns the slab sink's contents (or None)."""
        import contextlib

        from pfrl_amd.nn.mfma_linear import slab_sink

        with (slab_sink() if defer else contextlib.nullcontext()) as slabs:
            if all(l.dim() == 0 and l.dtype == torch.float32 for l in losses):
                one = self._unit_grad(losses[0])
                torch.autograd.backward(list(losses), [one] * len(losses))
            else:
                torch.autograd.backward(list(losses))
        return slabs

    def _step(self, loss, module, optimizer):
        optimizer.zero_grad()
        defer = self._defer_slabs([(module, optimizer)])
        slabs = self._backward([loss], defer)
        if module in self._reducers:
            self._reducers[module].all_reduce()
        if self.max_grad_norm is not None:
            clip_l2_grad_norm_(module.parameters(), self.max_grad_norm)
        if defer:
            optimizer.step(slabs=slabs)
        else:
            optimizer.step()

    def _unit_grad(self, loss):
        return _sac_losses.unit_grad(loss.device)

    def _soft_update_rides(self):
        """{critic parameter data_ptr: target tensor} when the soft target update may ride in the
        critics' optimizer launch: the stock ``sync_target_network`` (nobody overrode it), networks
        without buffers, parameter for parameter the same shapes.  None: sync_target_network()."""
        if (self.device.type != "cuda" or os.environ.get("PFRL_SAC_RIDERS", "1") == "0"
                or type(self).sync_target_network is not SoftActorCritic.sync_target_network):
            return None
        soft = {}
        for q, tq in ((self.q_func1, self.target_q_func1), (self.q_func2, self.target_q_func2)):
            ps, ts = list(q.parameters()), list(tq.parameters())
            if (next(q.buffers(), None) is not None or next(tq.buffers(), None) is not None
                    or len(ps) != len(ts) or any(a.shape != b.shape for a, b in zip(ps, ts))):
                return None
            for a, b in zip(ps, ts):
                soft[a.data_ptr()] = b.data
        return soft

    def _step_pair(self, loss1, loss2):
        self.q_func1_optimizer.zero_grad()
        self.q_func2_optimizer.zero_grad()
        pairs = [(self.q_func1, self.q_func1_optimizer), (self.q_func2, self.q_func2_optimizer)]
        defer = self._defer_slabs(pairs)
        slabs = self._backward([loss1, loss2], defer)
        for module in (self.q_func1, self.q_func2):
            if module in self._reducers:
                self._reducers[module].all_reduce()
            if self.max_grad_norm is not None:
                clip_l2_grad_norm_(module.parameters(), self.max_grad_norm)
        from pfrl_amd.optimizers import FusedAdam

        # (one launch for both when they are FusedAdam with equal hyperparameters; the gradient
        # slabs are summed and the target networks soft-updated in the same launch)
        soft = self._soft_update_rides()
        self._soft_done = bool(FusedAdam.step_together(
            [self.q_func1_optimizer, self.q_func2_optimizer], slabs=slabs, soft=soft,
            tau=self.soft_update_tau))

    @staticmethod
    def _q_pair(q1, q2, inputs):
        """(q1(inputs), q2(inputs), twinned): both networks as one chain of launches when they
        are the accelerated twin MLPs (pfrl_amd/nn/twin_mlp.py), else one after the other."""
        from pfrl_amd.nn.twin_mlp import twin_forward

        out = twin_forward(q1, q2, inputs)
        if out is None:
            return q1(inputs), q2(inputs), False
        return out[0], out[1], True

    def update_q_func(self, batch):
        batch_next_state = batch["next_state"]
        with torch.no_grad(), evaluating(self.policy), evaluating(self.target_q_func1), \
                evaluating(self.target_q_func2):
            next_actions, next_log_prob, _, _ = self._sample_policy(batch_next_state, False)
            next_q1, next_q2, _ = self._q_pair(self.target_q_func1, self.target_q_func2,
                                               (batch_next_state, next_actions))
            target_q = _sac_losses.soft_target_q(
                batch["reward"], batch["discount"], batch["is_state_terminal"], next_q1, next_q2,
                next_log_prob, self._loss_temperature())
        predict_q1, predict_q2, twinned = self._q_pair(self.q_func1, self.q_func2,
                                                       (batch["state"], batch["action"]))
        predict_q1, predict_q2 = torch.flatten(predict_q1), torch.flatten(predict_q2)
        if twinned:
            # The two critics share one autograd node: one backward pass for both losses, then
            # the two optimizer steps.  Same gradients and parameters as the reference's
            # q1-then-q2 order: neither loss depends on the other network, and both
            # predictions were computed before either step there too (:241-262).
            loss1, loss2 = _sac_losses.half_mse_pair(target_q, predict_q1, predict_q2)
            self._stat(q1=predict_q1, q2=predict_q2, loss1=loss1, loss2=loss2)
            self._step_pair(loss1, loss2)
            return
        loss1 = _sac_losses.half_mse(target_q, predict_q1)
        loss2 = _sac_losses.half_mse(target_q, predict_q2)
        self._stat(q1=predict_q1, q2=predict_q2, loss1=loss1, loss2=loss2)
        self._step(loss1, self.q_func1, self.q_func1_optimizer)
        self._step(loss2, self.q_func2, self.q_func2_optimizer)

    def update_temperature(self, log_prob):
        assert not log_prob.requires_grad
        from pfrl_amd import distributed

        if (distributed.world_size() == 1 and self.max_grad_norm is None
                and os.environ.get("PFRL_SAC_RIDERS", "1") != "0"
                and _sac_losses.temperature_step(self.temperature_holder, log_prob, self.entropy_target,
                                                 self.temperature_optimizer) is not None):
            return      # (loss and Adam step of the scalar in one launch)
        loss = _sac_losses.temperature_loss(self.temperature_holder, log_prob, self.entropy_target)
        if (isinstance(loss.grad_fn, _sac_losses._TemperatureLoss._backward_cls)
                and distributed.world_size() == 1 and self.max_grad_norm is None):
            # d loss / d log T is the loss itself (-mean(exp(log T) c)): hand it to the
            # optimizer without an autograd pass
            self.temperature_optimizer.zero_grad()
            p = self.temperature_holder.log_temperature
            p.grad = loss.detach().reshape(p.shape)
            self.temperature_optimizer.step()
            return
        self._step(loss, self.temperature_holder, self.temperature_optimizer)

    def update_policy_and_temperature(self, batch):
        batch_state = batch["state"]
        actions, log_prob, neg_log_prob, action_distrib = self._sample_policy(batch_state, True)
        # The policy loss needs dQ/da only.  With the Q parameters' requires_grad off while
        # this graph is recorded, backward skips their weight gradients, which the reference
        # computes, accumulates into q_func*.grad and never reads (the next update_q_func
        # starts with zero_grad): the parameters and every loss are unchanged.
        with _frozen(self.q_func1, self.q_func2):
            q1, q2, _ = self._q_pair(self.q_func1, self.q_func2, (batch_state, actions))
        loss = _sac_losses.policy_loss(log_prob, q1, q2, self._loss_temperature())
        self._step(loss, self.policy, self.policy_optimizer)
        if self.entropy_target is not None:
            self.update_temperature(log_prob.detach())
        with torch.no_grad():
            try:
                if action_distrib is None:       # (the fused head: a TransformedDistribution)
                    raise NotImplementedError
                ent = action_distrib.entropy()
            except NotImplementedError:
                # (the fused sample wrote -log_prob alongside log_prob)
                ent = neg_log_prob if neg_log_prob is not None else -log_prob
        self._stat(entropy=ent, policy_loss=loss)

    def _update_impl(self, batch, variant=None):
        self._soft_done = False
        self.update_q_func(batch)
        self.update_policy_and_temperature(batch)
        if not self._soft_done:     # (else it rode in the critics' optimizer launch, _step_pair)
            self.sync_target_network()

    def _after_update(self, variant=None):
        self.n_policy_updates += 1

    # -- acting (no explorer: the stochastic policy explores) -----------------------------------
    def batch_select_greedy_action(self, batch_obs, deterministic=False):
        with torch.no_grad(), evaluating(self.policy):
            batch_xs = self.batch_states(batch_obs, self.device, self.phi)
            policy_out = self.policy(batch_xs)
            if deterministic:
                return mode_of_distribution(policy_out).cpu().numpy()
            return policy_out.sample().cpu().numpy()

    def batch_act(self, batch_obs):
        if not self.training:
            return self.batch_select_greedy_action(batch_obs,
                                                   deterministic=self.act_deterministically)
        if self.burnin_action_func is not None and self.n_policy_updates == 0:
            batch_action = [self.burnin_action_func() for _ in range(len(batch_obs))]
        else:
            batch_action = self.batch_select_greedy_action(batch_obs)
        self.batch_last_obs = list(batch_obs)
        self.batch_last_action = list(batch_action)
        return batch_action

    def get_statistics(self):
        return [
            ("average_q1", self._mean_stat("q1")),
            ("average_q2", self._mean_stat("q2")),
            ("average_q_func1_loss", self._mean_stat("loss1")),
            ("average_q_func2_loss", self._mean_stat("loss2")),
            ("n_updates", self.n_policy_updates),
            ("average_entropy", self._mean_stat("entropy")),
            ("temperature", self.temperature),
        ]
