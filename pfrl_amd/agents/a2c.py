"""Synchronous advantage actor-critic on the device rollout path.

Mirrors ``pfrl.agents.a2c.A2C`` (/root/reference/pfrl/agents/a2c.py):
constructor (:53-73), ``_batch_act_train`` / ``_batch_observe_train``
(:224-287), ``_compute_returns`` (:150-167) and ``update`` (:169-213).

The reference already keeps its (T+1, N) rollout as torch tensors but stores
every observation as an fp32 copy and evaluates the return recursion as T
sequential tiny torch kernels.  Here observations stay as frame slots in a
DeviceFrameStore (one u8 write per frame), the update batch is produced by one
gather launch, and the GAE / n-step return recursion is one kernel
(pfrl_a2c_returns: one lane per env, reverse scan over T).
"""
import warnings
from logging import getLogger

import numpy as np
import torch

from pfrl_amd import agent, ops
from pfrl_amd.device_store import DeviceObsBatch
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_
from pfrl_amd.utils.mode_of_distribution import mode_of_distribution

logger = getLogger(__name__)


class A2C(agent.AttributeSavingMixin, agent.BatchAgent):
    """A2C: Advantage Actor-Critic (arguments as in the reference)."""

    process_idx = None
    saved_attributes = ("model", "optimizer")

    def __init__(self, model, optimizer, gamma, num_processes, gpu=None, update_steps=5,
                 phi=lambda x: x, pi_loss_coef=1.0, v_loss_coef=0.5, entropy_coeff=0.01,
                 use_gae=False, tau=0.95, act_deterministically=False, max_grad_norm=None,
                 average_actor_loss_decay=0.999, average_entropy_decay=0.999,
                 average_value_decay=0.999, batch_states=batch_states):
        self.model = model
        # With a GPU: observations are frame slots in HBM, returns come from pfrl_a2c_returns
        # (a missing library raises).  Without one (gpu=None / -1): the plumbing path -- the
        # (T+1, N, ...) observation tensor is collated on the host and the return scan is the
        # reference's T-step torch loop.
        self._on_gpu = gpu is not None and gpu >= 0
        if self._on_gpu:
            assert torch.cuda.is_available()
            self.device = torch.device("cuda:{}".format(gpu))
            self.model.to(self.device)
            from pfrl_amd import _native

            _native.lib()
        else:
            self.device = torch.device("cpu")
        self.optimizer = optimizer
        self.update_steps = update_steps
        self.num_processes = num_processes
        self.gamma = gamma
        self.use_gae = use_gae
        self.tau = tau
        self.act_deterministically = act_deterministically
        self.max_grad_norm = max_grad_norm
        self.phi = phi
        self.pi_loss_coef = pi_loss_coef
        self.v_loss_coef = v_loss_coef
        self.entropy_coeff = entropy_coeff
        self.average_actor_loss_decay = average_actor_loss_decay
        self.average_value_decay = average_value_decay
        self.average_entropy_decay = average_entropy_decay
        self.batch_states = batch_states
        self.t = 0
        self.t_start = 0
        self._avg = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.ingest = None
        self.frames = None
        self.refs = None          # int32 [T+1, N, k] on the device
        from pfrl_amd.distributed import GradientAllReducer

        self.grad_reducer = GradientAllReducer(self.model)
        if self._on_gpu:
            from pfrl_amd.staging import StagingRing

            self._stage = StagingRing(self.device, slot_bytes=1 << 20, n_slots=32)

    # -- observations (shared with PPO) ----------------------------------------
    def _refs_of(self, batch_obs):
        if isinstance(batch_obs, DeviceObsBatch):
            if self.frames is None:
                self.frames = batch_obs.store
                self._sample_obs = batch_obs[0]
            return batch_obs.refs
        if self.ingest is None:
            from pfrl_amd.replay_buffers.device_replay import DeviceReplayStore

            self.ingest = DeviceReplayStore(
                self.device, capacity=(self.update_steps + 2) * len(batch_obs) + 64, num_steps=1)
            self.ingest.set_phi(self.phi)
        pairs = [self.ingest.ingest(o) for o in batch_obs]
        self.ingest.flush()
        self.frames = self.ingest.frames
        return np.stack([p[0] for p in pairs]).astype(np.int32)

    def _divisor(self):
        if self.ingest is not None:
            return self.ingest.divisor_for(self.phi)
        from pfrl_amd.utils.batch_states import _divisor_for

        d = _divisor_for(self.phi, lambda: self._sample_obs.to_numpy())
        if d is None:
            raise TypeError("pfrl_amd.A2C: phi must be a cast/scale feature extractor")
        return d

    def _gather(self, refs_dev):
        if not self._on_gpu:
            return refs_dev
        x = self.frames.gather(refs_dev, self._divisor())
        fs = self.frames.frame_shape
        if refs_dev.shape[1] == 1:
            return x.view((x.shape[0],) + fs)
        if len(fs) >= 2 and fs[0] == 1:
            return x.view((x.shape[0], refs_dev.shape[1]) + fs[1:])
        return x

    def _sample_action(self, pout):
        return pout.sample()

    def _observe(self, batch_obs):
        """What the rollout stores per step and what ``_gather`` turns into the network input:
        frame-slot refs [N, k] on the device path, the collated batch itself on the host."""
        if not self._on_gpu:
            return self.batch_states(batch_obs, self.device, self.phi)
        (refs_dev,) = self._stage.upload([self._refs_of(batch_obs)])
        return refs_dev

    def _flush_storage(self, n_env, k, action):
        T, dev = self.update_steps, self.device
        self.action_shape = tuple(action.shape[1:])
        if self._on_gpu:
            self.refs = torch.zeros((T + 1, n_env, k), dtype=torch.int32, device=dev)
        else:       # k is the observation shape here
            self.refs = torch.zeros((T + 1, n_env) + tuple(k), dtype=torch.float, device=dev)
        self.actions = torch.zeros((T, n_env) + self.action_shape, dtype=torch.float, device=dev)
        self.rewards = torch.zeros((T, n_env), dtype=torch.float, device=dev)
        self.value_preds = torch.zeros((T + 1, n_env), dtype=torch.float, device=dev)
        self.returns = torch.zeros((T + 1, n_env), dtype=torch.float, device=dev)
        self.masks = torch.ones((T, n_env), dtype=torch.float, device=dev)

    # -- learning -----------------------------------------------------------------
    def _compute_returns(self, next_value):
        """reference :150-167 as one kernel launch."""
        if self.use_gae:
            self.value_preds[-1] = next_value
        else:
            self.returns[-1] = next_value
        if self._on_gpu:
            ops.a2c_returns(self.rewards, self.masks, self.value_preds, self.returns, self.gamma,
                            self.tau, self.use_gae)
            return
        running = 0
        for i in reversed(range(self.update_steps)):
            if self.use_gae:
                delta = (self.rewards[i] + self.gamma * self.value_preds[i + 1] * self.masks[i]
                         - self.value_preds[i])
                running = delta + self.gamma * self.tau * self.masks[i] * running
                self.returns[i] = running + self.value_preds[i]
            else:
                self.returns[i] = self.rewards[i] + self.gamma * self.returns[i + 1] * self.masks[i]

    def update(self):
        T, N = self.update_steps, self.num_processes
        with torch.no_grad():
            _, next_value = self.model(self._gather(self.refs[-1]))
            next_value = next_value[:, 0]
        self._compute_returns(next_value)
        pout, values = self.model(self._gather(
            self.refs[:-1].reshape((T * N,) + tuple(self.refs.shape[2:]))))
        actions = self.actions.reshape(-1, *self.action_shape)
        dist_entropy = pout.entropy().mean()
        action_log_probs = pout.log_prob(actions)
        values = values.reshape((T, N))
        action_log_probs = action_log_probs.reshape((T, N))
        advantages = self.returns[:-1] - values
        value_loss = (advantages * advantages).mean()
        action_loss = -(advantages.detach() * action_log_probs).mean()
        self.optimizer.zero_grad()
        (value_loss * self.v_loss_coef + action_loss * self.pi_loss_coef
         - dist_entropy * self.entropy_coeff).backward()
        self.grad_reducer.all_reduce()
        if self.max_grad_norm is not None:
            clip_l2_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        self.refs[0] = self.refs[-1]
        self.t_start = self.t
        self._last_losses = (value_loss.detach(), action_loss.detach(), dist_entropy.detach())
        # exponential moving statistics stay on the device (reference :200-213
        # pulls three scalars to the host every update)
        with torch.no_grad():
            x = torch.stack([action_loss.detach(), value_loss.detach(), dist_entropy.detach()])
            decay = torch.tensor([self.average_actor_loss_decay, self.average_value_decay,
                                  self.average_entropy_decay], device=self.device)
            self._avg += (1 - decay) * (x - self._avg)

    # -- acting ---------------------------------------------------------------------
    def batch_act(self, batch_obs):
        if self.training:
            return self._batch_act_train(batch_obs)
        return self._batch_act_eval(batch_obs)

    def batch_observe(self, batch_obs, batch_reward, batch_done, batch_reset):
        if self.training:
            self._batch_observe_train(batch_obs, batch_reward, batch_done, batch_reset)

    def _batch_act_train(self, batch_obs):
        assert self.training
        refs_dev = self._observe(batch_obs)
        statevar = self._gather(refs_dev)
        if self.t == 0:
            with torch.no_grad():
                pout, _ = self.model(statevar)
                action = pout.sample()   # shape probe; the reference draws here too (:231-234)
            self._flush_storage(refs_dev.shape[0], refs_dev.shape[1] if self._on_gpu
                                else refs_dev.shape[1:], action)
        self.refs[self.t - self.t_start] = refs_dev
        if self.t - self.t_start == self.update_steps:
            self.update()
        with torch.no_grad():
            pout, value = self.model(statevar)
            action = self._sample_action(pout)
        self.actions[self.t - self.t_start] = action.reshape(-1, *self.action_shape)
        self.value_preds[self.t - self.t_start] = value[:, 0]
        return action.cpu().numpy()

    def _batch_act_eval(self, batch_obs):
        assert not self.training
        with torch.no_grad():
            pout, _ = self.model(self._gather(self._observe(batch_obs)))
            action = mode_of_distribution(pout) if self.act_deterministically else pout.sample()
        return action.cpu().numpy()

    def _batch_observe_train(self, batch_obs, batch_reward, batch_done, batch_reset):
        assert self.training
        self.t += 1
        if any(batch_reset):
            warnings.warn(
                "A2C currently does not support resetting an env without reaching a"
                " terminal state during training. When receiving True in batch_reset,"
                " A2C considers it as True in batch_done instead.")
            batch_done = [bool(d) or bool(r) for d, r in zip(batch_done, batch_reset)]
        masks = np.array([0.0 if d else 1.0 for d in batch_done], dtype=np.float32)
        rewards = np.asarray(batch_reward, dtype=np.float32)
        if self._on_gpu:
            refs_dev, masks_dev, rewards_dev = self._stage.upload(
                [self._refs_of(batch_obs), masks, rewards])
        else:
            refs_dev = self._observe(batch_obs)
            masks_dev, rewards_dev = torch.from_numpy(masks), torch.from_numpy(rewards)
        i = self.t - self.t_start
        self.masks[i - 1] = masks_dev
        self.rewards[i - 1] = rewards_dev
        self.refs[i] = refs_dev
        if i == self.update_steps:
            self.update()

    def get_statistics(self):
        a = self._avg.cpu().numpy()
        return [("average_actor", float(a[0])), ("average_value", float(a[1])),
                ("average_entropy", float(a[2]))]
