"""Deep Q-Network on the device-resident replay path.

Mirrors ``pfrl.agents.dqn.DQN`` (/root/reference/pfrl/agents/dqn.py): same
constructor (:181-206), ``batch_act`` (:490-507), ``batch_observe`` with the
per-env append -> update interleaving (:509-549), ``update`` (:316-365), the
Huber / MSE losses (:44-104), target synchronisation (:307-314), statistics
(:812-819) and snapshots (:794-810).  Differences are in *where* data lives,
not in what is computed:

* minibatches come from ``batch_experiences`` = one fused HIP launch over the
  HBM replay store (no per-update H2D of fp32 stacks);
* TD errors for prioritized replay are handed to ``update_errors`` as a device
  tensor (the reference does ``.cpu().numpy()``, dqn.py:449-454);
* ``q_record`` / ``loss_record`` keep device tensors and are reduced only when
  ``get_statistics()`` is called (the reference blocks on a D2H copy per update,
  dqn.py:358,445);
* with ``torch.distributed`` initialised, gradients are all-reduced (RCCL over
  xGMI) before the optimizer step -- env-sharded data parallelism.

``recurrent=True`` (reference :232-241, :367-386, :472-488) replays whole episodes from an
episodic buffer as packed sequences; shapes vary per update, so that mode runs eagerly on stock
torch ops (no graphs, no fused gathers) with the episodes on the host.  Every model call of a
loss goes through ``_action_value`` so that the DQN-family subclasses inherit it.
"""
import collections
import copy
import os
from logging import getLogger

import numpy as np
import torch
import torch.nn.functional as F

from pfrl_amd import agent
from pfrl_amd.replay_buffer import (AbstractEpisodicReplayBuffer, DeviceExperienceBatch,
                                    ReplayUpdater, batch_experiences,
                                    batch_recurrent_experiences)
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.clip_l2_grad_norm import clip_l2_grad_norm_
from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.copy_param import synchronize_parameters
from pfrl_amd.utils.recurrent import (get_recurrent_state_at, mask_recurrent_state_at,
                                      one_step_forward, pack_and_forward,
                                      recurrent_state_as_numpy)


class _Pending:
    def __repr__(self):
        return "greedy(pending)"


_PENDING = _Pending()


def _pending_greedy():
    return _PENDING


def _mean_or_nan(xs):
    return float(np.mean(xs)) if len(xs) else np.nan


def compute_value_loss(y, t, clip_delta=True, batch_accumulator="mean"):
    """Huber(delta=1) or half-MSE value loss, 'mean' or 'sum' over the batch
    (reference :44-68)."""
    assert batch_accumulator in ("mean", "sum")
    y = y.reshape(-1, 1)
    t = t.reshape(-1, 1)
    if clip_delta:
        return F.smooth_l1_loss(y, t, reduction=batch_accumulator)
    return F.mse_loss(y, t, reduction=batch_accumulator) / 2


def compute_weighted_value_loss(y, t, weights, clip_delta=True, batch_accumulator="mean"):
    """Importance-weighted variant (reference :71-104)."""
    assert batch_accumulator in ("mean", "sum")
    y = y.reshape(-1, 1)
    t = t.reshape(-1, 1)
    if clip_delta:
        losses = F.smooth_l1_loss(y, t, reduction="none")
    else:
        losses = F.mse_loss(y, t, reduction="none") / 2
    losses = losses.reshape(-1)
    loss_sum = torch.sum(losses * weights.to(losses.device))
    if batch_accumulator == "mean":
        return loss_sum / y.shape[0]
    return loss_sum


def make_target_model_as_copy(model):
    target_model = copy.deepcopy(model)
    target_model.eval()
    return target_model


class _DeviceRecord:
    """Bounded record of the most recent ``maxlen`` scalars kept as device
    tensors; reduced on demand."""

    def __init__(self, maxlen):
        self.maxlen = maxlen
        self._chunks = collections.deque()
        self._count = 0

    def extend(self, t):
        t = t.detach().reshape(-1)
        self._chunks.append(t)
        self._count += t.numel()
        while self._count - self._chunks[0].numel() >= self.maxlen:
            self._count -= self._chunks.popleft().numel()

    def values(self):
        if not self._chunks:
            return np.zeros(0, dtype=np.float32)
        allv = torch.cat([c.float() for c in self._chunks]).cpu().numpy()
        return allv[-self.maxlen:]

    def __len__(self):
        return min(self._count, self.maxlen)


class DQN(agent.AttributeSavingMixin, agent.BatchAgent):
    """Deep Q-Network algorithm (see module docstring for the argument list;
    it is the reference's)."""

    saved_attributes = ("model", "target_model", "optimizer")
    _fused_td_double = False
    # recurrent=True: every model call of a loss goes through _action_value (subclasses too)

    def __init__(self, q_function, optimizer, replay_buffer, gamma, explorer, gpu=None,
                 replay_start_size=50000, minibatch_size=32, update_interval=1,
                 target_update_interval=10000, clip_delta=True, phi=lambda x: x,
                 target_update_method="hard", soft_update_tau=1e-2, n_times_update=1,
                 batch_accumulator="mean", episodic_update_len=None, logger=getLogger(__name__),
                 batch_states=batch_states, recurrent=False, max_grad_norm=None,
                 use_graphs=None, step_fused_gather=None, batch_target_pass=None,
                 fused_td_loss=True, replay_overlap=None, step_fused_chunks=(0.1, 0.4)):
        self.model = q_function
        if gpu is not None and gpu >= 0:
            assert torch.cuda.is_available()
            self.device = torch.device("cuda:{}".format(gpu))
            self.model.to(self.device)
            # narrow output layers (Linear(512, n_actions)) take the one-launch head kernels;
            # same module class, parameters and state_dict
            from pfrl_amd.nn.mfma_trunk import accelerate_heads

            accelerate_heads(self.model)
        else:
            self.device = torch.device("cpu")
        self.recurrent = bool(recurrent)
        if self.recurrent:
            # DRQN (reference :232-241): whole episodes are replayed as packed sequences, so
            # shapes vary from update to update -- this mode runs eagerly on stock torch ops.
            assert isinstance(replay_buffer, AbstractEpisodicReplayBuffer)
            use_graphs = step_fused_gather = fused_td_loss = False
        self.replay_buffer = replay_buffer
        if hasattr(replay_buffer, "bind"):
            replay_buffer.bind(self.device, phi)
        self.optimizer = optimizer
        self.gamma = gamma
        self.explorer = explorer
        self.gpu = gpu
        self.target_update_interval = target_update_interval
        self.clip_delta = clip_delta
        self.phi = phi
        self.target_update_method = target_update_method
        self.soft_update_tau = soft_update_tau
        self.batch_accumulator = batch_accumulator
        assert batch_accumulator in ("mean", "sum")
        self.logger = logger
        self.batch_states = batch_states
        self.replay_updater = ReplayUpdater(
            replay_buffer=replay_buffer,
            update_func=self.update_from_episodes if self.recurrent else self.update,
            batchsize=minibatch_size,
            episodic_update=self.recurrent, episodic_update_len=episodic_update_len,
            n_times_update=n_times_update, replay_start_size=replay_start_size,
            update_interval=update_interval)
        self.minibatch_size = minibatch_size
        self.episodic_update_len = episodic_update_len
        self.replay_start_size = replay_start_size
        self.update_interval = update_interval
        self.max_grad_norm = max_grad_norm
        assert target_update_interval % update_interval == 0, \
            "target_update_interval should be a multiple of update_interval"
        self.t = 0
        self.optim_t = 0
        self._cumulative_steps = 0
        self.target_model = make_target_model_as_copy(self.model)
        self.q_record = _DeviceRecord(1000)
        self.loss_record = _DeviceRecord(100)
        self.batch_last_obs = []
        self.batch_last_action = []
        # recurrent states of the model (reference :266-269)
        self.train_recurrent_states = None
        self.train_prev_recurrent_states = None
        self.test_recurrent_states = None
        if (self.replay_buffer.capacity is not None
                and self.replay_buffer.capacity < self.replay_updater.replay_start_size):
            raise ValueError("Replay start size cannot exceed replay buffer capacity.")
        # data-parallel gradient averaging (no-op for a single process)
        from pfrl_amd.distributed import GradientAllReducer

        self.grad_reducer = GradientAllReducer(self.model)
        # HIP-graph replay of the update and one fused gather per env step are
        # device-path optimisations; both compute exactly what eager mode does.
        on_gpu = self.device.type == "cuda"
        self.use_graphs = on_gpu if use_graphs is None else bool(use_graphs and on_gpu)
        self.step_fused_gather = on_gpu if step_fused_gather is None else \
            bool(step_fused_gather and on_gpu)
        # one target-network forward for all minibatches of an env step (the target
        # network is frozen between syncs, so the values are the same function of
        # the same inputs; only the batch size of the conv/GEMM kernels changes)
        self.batch_target_pass = self.step_fused_gather if batch_target_pass is None else \
            bool(batch_target_pass and self.step_fused_gather)
        self.fused_td_loss = bool(fused_td_loss)
        # env-range boundaries (fractions of num_envs) of the step-fused path
        self.step_fused_chunks = tuple(step_fused_chunks)
        self._chunks_set_by_caller = tuple(step_fused_chunks) != (0.1, 0.4)
        self._target_raw_bufs = {}
        self._single_bufs = {}
        self.range_graphs = os.environ.get("PFRL_RANGE_GRAPHS", "1") != "0"
        # no-host-round-trip step for device envs (agents/_dqn_device_step.py)
        self.device_step = (on_gpu and self.step_fused_gather
                            and os.environ.get("PFRL_DEVICE_STEP", "1") != "0")
        self._last_actions_dev = None
        self._analytic_backward = None
        self._graphed = None
        self._last_y = None
        # Prioritized replay is a serial chain per update (priorities -> B dependent
        # draws -> gather) that the next forward pass has to wait for, but the
        # backward pass and the optimizer step do not: with graphs on, the buffer's
        # launches move to their own HIP stream and overlap them (see
        # _update_from_batch).  Same launches, same order within the replay stream.
        self._replay_stream = None
        self._fwd_event = None
        from pfrl_amd.replay_buffers.prioritized import PrioritizedReplayBuffer

        want_overlap = (self.use_graphs and isinstance(replay_buffer, PrioritizedReplayBuffer)
                        and getattr(replay_buffer, "is_device", False))
        if want_overlap if replay_overlap is None else (replay_overlap and want_overlap):
            self._replay_stream = torch.cuda.Stream(self.device)
            self._fwd_event = torch.cuda.Event()
            self._replay_stream.wait_stream(torch.cuda.current_stream(self.device))
            replay_buffer.set_replay_stream(self._replay_stream)

    @property
    def cumulative_steps(self):
        return self._cumulative_steps

    def sync_target_network(self):
        self._flush_backward()           # (the parameters of the last update, not the one before)
        synchronize_parameters(src=self.model, dst=self.target_model,
                               method=self.target_update_method, tau=self.soft_update_tau)
        self._side_needs_main = True     # (a side stream reading the target network: see _range_side)

    _side_needs_main = True

    # -- learning -------------------------------------------------------------
    def update(self, experiences, errors_out=None, _gathered=None):
        """One minibatch update (reference :316-365).  ``_gathered``: the result of
        ``batch_experiences`` when the caller has launched the gather already."""
        # (a held-back backward + step of the previous minibatch goes first: ahead of the wait
        # for this minibatch's gather, which it does not depend on)
        self._flush_backward()
        if isinstance(experiences, DeviceExperienceBatch):
            has_weight = experiences.has_weight
        else:
            has_weight = "weight" in experiences[0][0]
        exp_batch = _gathered if _gathered is not None else batch_experiences(
            experiences, device=self.device, phi=self.phi, gamma=self.gamma,
            batch_states=self.batch_states)
        if self._replay_stream is not None and isinstance(experiences, DeviceExperienceBatch):
            if self._graphed is not None and self.use_graphs and has_weight:
                # NoisyNet draws of this update (one launch): ahead of the wait, not behind it
                self._graphed.prefill(exp_batch, True)
            # the minibatch was sampled and gathered on the replay stream
            torch.cuda.current_stream(self.device).wait_event(experiences.store.ready_event)
        if has_weight and "weights" not in exp_batch:
            exp_batch["weights"] = torch.tensor([e[0]["weight"] for e in experiences],
                                                device=self.device, dtype=torch.float32)
        self._update_from_batch(exp_batch, has_weight, errors_out)

    def update_from_episodes(self, episodes, errors_out=None):
        """One update from sampled episodes, longest first (reference :367-386)."""
        assert errors_out is None, "Recurrent DQN does not support PrioritizedBuffer"
        episodes = sorted(episodes, key=len, reverse=True)
        exp_batch = batch_recurrent_experiences(episodes, device=self.device, phi=self.phi,
                                                gamma=self.gamma, batch_states=self.batch_states)
        self._update_from_batch(exp_batch)

    def _update_from_batch(self, exp_batch, has_weight=False, errors_out=None, deferred=None):
        """``deferred``: a list that receives this update's (loss, y) graph outputs
        instead of cloning them now -- valid when every update of the step replays
        its own graph (step-fused path), so the outputs stay intact until the end
        of the step and are recorded with one concatenation."""
        want_errors = has_weight or errors_out is not None
        handed_over = []
        self._flush_backward()
        if self.use_graphs:
            hand_over = None
            late = None
            if has_weight and errors_out is None and self._replay_stream is not None:
                if self._late_backward_ok():
                    late = self._late
                def hand_over(delta):
                    # TD errors exist as soon as the forward graph has run: the replay
                    # stream takes them (priority update, then the next sample and
                    # gather) while backward + optimizer step continue on this stream
                    self._fwd_event.record(torch.cuda.current_stream(self.device))
                    self._replay_stream.wait_event(self._fwd_event)
                    self.replay_buffer.update_errors(delta)
                    handed_over.append(True)
            loss, delta = self._graphed_step(exp_batch, want_errors, deferred, hand_over, late)
        else:
            loss, delta = self._compute_loss(exp_batch, want_errors=want_errors)
        if errors_out is not None:
            del errors_out[:]
            errors_out.extend(delta.cpu().numpy())
        if has_weight and not handed_over:
            if self._replay_stream is not None:
                self.replay_buffer.replay_stream_wait_current()
            self.replay_buffer.update_errors(delta if errors_out is None else errors_out)
        if not self.use_graphs:
            self.loss_record.extend(loss)
            self.optimizer.zero_grad()
            loss.backward()
            self.grad_reducer.all_reduce()
            if self.max_grad_norm is not None:
                clip_l2_grad_norm_(self.model.parameters(), self.max_grad_norm)
            self.optimizer.step()
        self.optim_t += 1

    # -- backward + step of update k launched after the replay side of update k + 1 -----------
    # With a replay stream the next forward pass waits for: priorities of this minibatch ->
    # pending appends -> the B dependent draws -> the gather.  That chain starts as soon as the
    # forward graph has produced the TD errors; the backward / optimizer graph only has to be
    # done by the time the gather is.  The host, however, launches in program order, and a
    # graph launch costs it ~90 us: launched first, the backward graph kept the chain waiting
    # for the host (measured with events, tools/pipeline_events.py: 60 us from the end of the
    # forward graph to the first replay-side kernel, 40 us of appends behind it).  So the
    # launch is held back: update() of the NEXT minibatch (whose sample() has just enqueued
    # the chain) replays it first thing, and so does everything else that touches the model.
    _late = None
    _in_train_step = False

    def _late_backward_ok(self):
        if self.__dict__.get("_late") is None:
            self._late = []
            self._late_enabled = os.environ.get("PFRL_LATE_BACKWARD", "1") != "0"
        # (only inside batch_observe, which flushes on its way out: a caller of update() itself
        # expects the parameters to be stepped when it returns)
        return (self._late_enabled and self._in_train_step
                and not getattr(self._graphed, "split_for_allreduce", False))

    def _flush_backward(self):
        late = self.__dict__.get("_late")
        if late:
            pending = list(late)
            del late[:]
            for replay_rest in pending:
                replay_rest()

    def _graphed_step(self, exp_batch, want_errors, deferred=None, after_forward=None, late=None):
        """loss -> backward -> step replayed from a captured HIP graph."""
        if self._graphed is None:
            from pfrl_amd.agents.graphed_update import GraphedUpdate

            self._graphed = GraphedUpdate(self)
            self._graphed.pipeline = self._replay_stream is not None
        try:
            loss, delta, y = self._graphed.run(exp_batch, want_errors, after_forward, late)
        except Exception as e:  # capture not possible -> stay eager on the GPU
            if self._graphed.graphs:
                raise
            self.logger.warning("HIP-graph capture of the update failed (%s); running eagerly", e)
            self.use_graphs = False
            loss, delta = self._compute_loss(exp_batch, want_errors=want_errors)
            self.loss_record.extend(loss)
            self.optimizer.zero_grad()
            loss.backward()
            self.grad_reducer.all_reduce()
            if self.max_grad_norm is not None:
                clip_l2_grad_norm_(self.model.parameters(), self.max_grad_norm)
            self.optimizer.step()
            return loss, delta
        # graph outputs are static buffers: copy what the records keep
        if deferred is not None:
            deferred.append((loss, y))
        else:
            self.loss_record.extend(loss.clone())
            self.q_record.extend(y.clone())
        return loss, delta

    # -- step-batched target pass ------------------------------------------------
    def _target_is_deterministic(self):
        """The frozen target network may be evaluated for all minibatches of an
        env step in one batch only if its forward pass draws no noise."""
        from pfrl_amd.nn.noisy_linear import FactorizedNoisyLinear

        return not any(isinstance(m, (FactorizedNoisyLinear, torch.nn.Dropout))
                       for m in self.target_model.modules())

    def _precompute_target_raw(self, next_states):
        """target_model over ALL next-states of the step (U*B observations) in one
        forward pass; returns the raw tensor the ActionValue is rebuilt from."""
        with torch.no_grad():
            av = self.target_model(next_states)
        if hasattr(av, "q_dist"):
            self._target_z_values = av.z_values
            return av.q_dist
        return av.q_values

    def _action_value(self, model, states, recurrent_state):
        """``model(states)``; for a recurrent model ``states`` is a list of per-episode batches
        and the result is flat in packed (time-major) order (reference :391-399, :415-420)."""
        if self.recurrent:
            return pack_and_forward(model, states, recurrent_state)[0]
        return model(states)

    def _target_next_action_value(self, exp_batch):
        raw = exp_batch.get("target_next_raw")
        if raw is None:
            return self._action_value(self.target_model, exp_batch["next_state"],
                                      exp_batch.get("next_recurrent_state"))
        if raw.ndim == 3:
            from pfrl_amd.action_value import DistributionalDiscreteActionValue

            return DistributionalDiscreteActionValue(raw, self._target_z_values)
        from pfrl_amd.action_value import DiscreteActionValue

        return DiscreteActionValue(raw)

    def _compute_target_values(self, exp_batch):
        target_next_qout = self._target_next_action_value(exp_batch)
        next_q_max = target_next_qout.max
        return (exp_batch["reward"]
                + exp_batch["discount"] * (1.0 - exp_batch["is_state_terminal"]) * next_q_max)

    def _compute_y_and_t(self, exp_batch):
        batch_size = exp_batch["reward"].shape[0]
        qout = self._action_value(self.model, exp_batch["state"], exp_batch.get("recurrent_state"))
        batch_q = torch.reshape(qout.evaluate_actions(exp_batch["action"]), (batch_size, 1))
        with torch.no_grad():
            batch_q_target = torch.reshape(self._compute_target_values(exp_batch), (batch_size, 1))
        return batch_q, batch_q_target

    # The fused TD-loss launch applies to DQN (_fused_td_double False) and
    # DoubleDQN (True); subclasses that override the target / loss computation
    # (categorical agents) set it to None and keep their own.

    def _fused_td_loss_applicable(self):
        """The fused launch computes the stock DQN target (``_fused_td_double`` False) or the
        stock Double-DQN target (True).  ``_compute_target_values`` / ``_compute_y_and_t`` /
        ``_compute_loss`` are the reference's extension points (its own DoubleDQN overrides
        the first): a subclass that overrides any of them keeps the composite path, on
        every device."""
        cls = type(self)
        if not (self.fused_td_loss and self.device.type == "cuda" and not self.recurrent
                and cls._fused_td_double is not None
                and cls._compute_y_and_t is DQN._compute_y_and_t
                and cls._compute_loss is DQN._compute_loss):
            return False
        if cls._fused_td_double:
            from pfrl_amd.agents.double_dqn import DoubleDQN

            return cls._compute_target_values is DoubleDQN._compute_target_values
        return cls._compute_target_values is DQN._compute_target_values

    # set by the graph path (GraphedUpdate), which runs backward right after the loss and
    # flushes the queue: the head's batch sums then ride on the trunk backward's fold launch
    _defer_head_fold = False

    def _head_split(self):
        """(body modules, head layer) when the model is ``nn.Sequential(..., Linear(K, A),
        DiscreteActionValueHead())`` with the narrow head on the GPU kernels (the example
        Q-functions): the head then runs inside the loss launch.  None otherwise, and under data
        parallelism (the gradient hooks of the early all-reduce hang on autograd's accumulation)."""
        cached = self.__dict__.get("_head_split_cache")
        if cached is not None and cached[0] is self.model:
            return cached[1]
        from pfrl_amd.nn.mfma_trunk import _SmallLinearSlot
        from pfrl_amd.q_functions import DiscreteActionValueHead

        out = None
        m = self.model
        if isinstance(m, torch.nn.Sequential) and len(m) >= 3:
            mods = list(m._modules.values())
            if (type(mods[-1]) is DiscreteActionValueHead and isinstance(mods[-2], _SmallLinearSlot)
                    and mods[-2].bias is not None
                    and os.environ.get("PFRL_FUSE_HEAD_LOSS", "1") != "0"):
                out = (mods[:-2], mods[-2])
        self._head_split_cache = (m, out)
        return out

    def _compute_loss_fused(self, exp_batch, errors_out, record):
        from pfrl_amd import distributed, ops

        # (data parallel: only inside a captured update whose optimizer takes the head's slabs --
        # they are then folded into the flat bucket; the eager path keeps autograd's head, whose
        # gradient hooks the early all-reduce hangs on)
        split = self._head_split() if (
            distributed.world_size() == 1
            or (self._defer_head_fold and getattr(self, "_graphed", None) is not None
                and self._graphed._optimizer_finishes_gradients())) else None
        qout = None
        h_fold = None
        if split is not None:
            from pfrl_amd.nn import mfma_trunk

            h = exp_batch["state"]
            # the hidden layer may leave its output as split-K slabs for the head launch to fold
            sink = mfma_trunk.FWD_FOLD_SINK = {} if (
                self._defer_head_fold and os.environ.get("PFRL_FWD_FOLD", "1") != "0") else None
            try:
                body = split[0]
                for k, mod in enumerate(body):
                    if sink and k > 0:
                        # slabs left by an earlier module are only legal as the HEAD's input: a module
                        # in between (e.g. an extra hidden layer) must see the folded tensor
                        mfma_trunk.flush_fwd_folds(sink)
                    h = mod(h)
            finally:
                mfma_trunk.FWD_FOLD_SINK = None
            head = split[1]
            usable = torch.is_tensor(h) and ops.dqn_head_td_loss_supported(h, head.weight, head.bias)
            if sink:
                rec = sink.pop(h.data_ptr(), None) if usable else None
                mfma_trunk.flush_fwd_folds(sink)        # (anything that is not the head's input)
                if rec is not None:
                    h_fold = (rec[1], rec[2], rec[3], rec[4])
            if not usable:
                # outside the fused launch (e.g. a head input that is not [B, 256 | 512]): finish
                # the forward pass from h, separate launches below
                qout = list(self.model._modules.values())[-1](head(h))
                split = None
        if split is not None:
            with torch.no_grad():
                target_q = self._target_next_action_value(exp_batch).q_values
                next_online = None
                if type(self)._fused_td_double:
                    with evaluating(self.model):
                        next_online = self.model(exp_batch["next_state"]).q_values
            loss, y, delta = ops.dqn_head_td_loss(
                h, head.weight, head.bias, exp_batch["action"], target_q, next_online,
                exp_batch["reward"], exp_batch["discount"], exp_batch["is_state_terminal"],
                exp_batch.get("weights"), self.clip_delta, self.batch_accumulator == "mean",
                defer=self._defer_head_fold, h_fold=h_fold)
            # the gradients came out of the same launch: backward may start at h, and the
            # head's own gradients are handed over as they are
            if loss.grad_fn is not None:
                dh, dw, db = loss.grad_fn.saved_tensors
                self._analytic_backward = (h, dh, [(head.weight, dw), (head.bias, db)])
            else:
                self._analytic_backward = None
            self._last_y = y
            if record:
                self.q_record.extend(y)
            if errors_out is not None:
                del errors_out[:]
                errors_out.extend(delta.cpu().numpy())
            return loss, delta
        if qout is None:
            qout = self.model(exp_batch["state"])
        with torch.no_grad():
            target_q = self._target_next_action_value(exp_batch).q_values
            next_online = None
            if type(self)._fused_td_double:
                with evaluating(self.model):
                    next_online = self.model(exp_batch["next_state"]).q_values
        loss, y, delta = ops.dqn_td_loss(
            qout.q_values, exp_batch["action"], target_q, next_online, exp_batch["reward"],
            exp_batch["discount"], exp_batch["is_state_terminal"], exp_batch.get("weights"),
            self.clip_delta, self.batch_accumulator == "mean")
        # d(loss)/dQ(s) is already known: callers may start backward at Q(s) and skip
        # the ones-fill and the multiply autograd would launch for loss.backward()
        self._analytic_backward = (qout.q_values, loss.grad_fn.saved_tensors[0]
                                   if loss.grad_fn is not None else None)
        self._last_y = y
        if record:
            self.q_record.extend(y)
        if errors_out is not None:
            del errors_out[:]
            errors_out.extend(delta.cpu().numpy())
        return loss, delta

    def _compute_loss(self, exp_batch, errors_out=None, want_errors=False, record=True):
        """Returns (loss, |y - t| per sample or None).  ``errors_out`` keeps the
        reference's list-filling form for callers that use it directly."""
        if self._fused_td_loss_applicable():
            return self._compute_loss_fused(exp_batch, errors_out, record)
        y, t = self._compute_y_and_t(exp_batch)
        self._last_y = y.detach()
        if record:
            self.q_record.extend(y)
        delta = None
        if errors_out is not None or want_errors:
            delta = torch.abs(y.detach() - t)
            if delta.ndim == 2:
                delta = torch.sum(delta, dim=1)
            if errors_out is not None:
                del errors_out[:]
                errors_out.extend(delta.cpu().numpy())
        if "weights" in exp_batch:
            loss = compute_weighted_value_loss(y, t, exp_batch["weights"],
                                               clip_delta=self.clip_delta,
                                               batch_accumulator=self.batch_accumulator)
        else:
            loss = compute_value_loss(y, t, clip_delta=self.clip_delta,
                                      batch_accumulator=self.batch_accumulator)
        return loss, delta

    # -- acting ---------------------------------------------------------------
    def _route_observation_layout(self, batch_obs=None):
        """A channels_last network gets its minibatches gathered straight into that
        memory format (stacks of four u8 planes); checked per call because the model
        may be converted after the agent was built."""
        from pfrl_amd.device_store import DeviceObs, DeviceObsBatch
        from pfrl_amd.nn.atari_cnn import wants_channels_last

        want = self.device.type == "cuda" and wants_channels_last(self.model)
        store = getattr(getattr(self.replay_buffer, "store", None), "frames", None)
        if store is not None:
            store.emit_channels_last = want
        if batch_obs is not None:
            first = batch_obs if isinstance(batch_obs, DeviceObsBatch) else (
                batch_obs[0] if len(batch_obs) else None)
            if isinstance(first, (DeviceObs, DeviceObsBatch)):
                first.store.emit_channels_last = want

    def _evaluate_model(self, batch_obs):
        self._route_observation_layout(batch_obs)
        batch_xs = self.batch_states(batch_obs, self.device, self.phi)
        if not self.recurrent:
            return self.model(batch_xs)
        # one step for every env; training keeps the state from before the step too, because
        # it is stored with the transition (reference :472-488)
        if self.training:
            self.train_prev_recurrent_states = self.train_recurrent_states
            batch_av, self.train_recurrent_states = one_step_forward(
                self.model, batch_xs, self.train_recurrent_states)
        else:
            batch_av, self.test_recurrent_states = one_step_forward(
                self.model, batch_xs, self.test_recurrent_states)
        return batch_av

    @staticmethod
    def _restart_ended(recurrent_states, batch_done, batch_reset):
        """Zero the state of every env whose episode ended (reference :107-128)."""
        ended = [i for i, (d, r) in enumerate(zip(batch_done, batch_reset)) if d or r]
        return mask_recurrent_state_at(recurrent_states, ended) if ended else recurrent_states

    def _host_batch_to_device(self, batch_obs):
        """Training observations that arrive as host LazyFrames (a real VectorFrameStack /
        MultiprocessVectorEnv) enter the replay buffer's frame ring HERE, once per batch:
        only the frames not seen before cross PCIe (one 7 KB frame per env and step instead of
        the 28 KB stack), identity is resolved for the whole batch in one pass, and from then
        on the batch is a DeviceObsBatch -- acting gathers it on the device and the step-fused
        path appends it with array writes, exactly as for a device env."""
        from pfrl_amd.device_store import DeviceObs, DeviceObsBatch

        store = getattr(self.replay_buffer, "store", None)
        if (store is None or self.recurrent or isinstance(batch_obs, DeviceObsBatch)
                or len(batch_obs) == 0 or isinstance(batch_obs[0], DeviceObs)):
            return batch_obs
        out = store.ingest_many(batch_obs)
        if out is None:
            return batch_obs
        return DeviceObsBatch(store.frames, out[0], out[1])

    def _explorer_draws_before_greedy(self):
        from pfrl_amd.explorers.epsilon_greedy import _EpsilonGreedyBase

        ex = self.explorer
        return (isinstance(ex, _EpsilonGreedyBase)
                and type(ex).select_action is _EpsilonGreedyBase.select_action
                and self.device.type == "cuda")

    def batch_act(self, batch_obs):
        if self._replay_stream is not None:
            # frames the buffer uploaded on its own stream must be visible to the gather
            self.replay_buffer.current_wait_replay_stream()
        if self.training:
            batch_obs = self._host_batch_to_device(batch_obs)
            if self.device_step:
                # device env + uniform device replay: draws by the native planner, actions stay
                # in HBM, no D2H inside the step (agents/_dqn_device_step.py)
                from pfrl_amd.agents import _dqn_device_step

                out = _dqn_device_step.act(self, batch_obs)
                if out is not None:
                    return out
            self._last_actions_dev = None
        with torch.no_grad(), evaluating(self.model):
            batch_av = self._evaluate_model(batch_obs)
            greedy_dev = batch_av.greedy_actions.detach()
        if self.training and self._explorer_draws_before_greedy():
            # epsilon-greedy family: the per-env draws (rand(), then random_action_func when it
            # fires -- same stream, same order as reference :494-502) do not depend on the
            # network, so they run while the Q forward pass is still on the GPU; the greedy
            # entries are filled in after the one D2H read
            select = self.explorer.select_action
            batch_action = [select(self.t, _pending_greedy, action_value=None)
                            for _ in range(len(batch_obs))]
            if any(a is _PENDING for a in batch_action):
                batch_argmax = greedy_dev.cpu().numpy()
                batch_action = [batch_argmax[i] if a is _PENDING else a
                                for i, a in enumerate(batch_action)]
            self.batch_last_obs = list(batch_obs)
            self.batch_last_action = list(batch_action)
            self._last_obs_batch = batch_obs
            return batch_action
        batch_argmax = greedy_dev.cpu().numpy()
        if self.training:
            select = self.explorer.select_action
            if getattr(self.explorer, "uses_action_value", True):
                # e.g. Boltzmann: every env gets its own row (reference :494-502)
                batch_action = [
                    select(self.t, lambda i=i: batch_argmax[i], action_value=batch_av[i:i + 1])
                    for i in range(len(batch_obs))
                ]
            else:
                batch_action = [
                    select(self.t, lambda i=i: batch_argmax[i], action_value=None)
                    for i in range(len(batch_obs))
                ]
            self.batch_last_obs = list(batch_obs)
            self.batch_last_action = list(batch_action)
            self._last_obs_batch = batch_obs
        else:
            batch_action = batch_argmax
        return batch_action

    def _append_transition(self, i, batch_obs, batch_reward, batch_done, batch_reset):
        rbuf = self.replay_buffer
        if self.batch_last_obs[i] is not None:
            assert self.batch_last_action[i] is not None
            extra = {}
            if self.recurrent:
                for key, states in (("recurrent_state", self.train_prev_recurrent_states),
                                    ("next_recurrent_state", self.train_recurrent_states)):
                    extra[key] = recurrent_state_as_numpy(
                        get_recurrent_state_at(states, i, detach=True))
            rbuf.append(state=self.batch_last_obs[i], action=self.batch_last_action[i],
                        reward=batch_reward[i], next_state=batch_obs[i], next_action=None,
                        is_state_terminal=batch_done[i], env_id=i, **extra)
            if batch_reset[i] or batch_done[i]:
                self.batch_last_obs[i] = None
                self.batch_last_action[i] = None
                rbuf.stop_current_episode(env_id=i)

    def _batch_observe_train(self, batch_obs, batch_reward, batch_done, batch_reset):
        rbuf = self.replay_buffer
        batch_obs = self._host_batch_to_device(batch_obs)
        if self._replay_stream is not None:
            # this env step's frames (written on the compute stream) before any gather
            rbuf.replay_stream_wait_current()
        if self.step_fused_gather and getattr(rbuf, "supports_lookahead", False):
            return self._batch_observe_train_fused(batch_obs, batch_reward, batch_done,
                                                   batch_reset)
        if self._per_lookahead_ok():
            return self._batch_observe_train_per(batch_obs, batch_reward, batch_done, batch_reset)
        updater = self.replay_updater
        for i in range(len(batch_obs)):
            self.t += 1
            self._cumulative_steps += 1
            if self.t % self.target_update_interval == 0:
                self.sync_target_network()
            self._append_transition(i, batch_obs, batch_reward, batch_done, batch_reset)
            updater.update_if_necessary(self.t)
        if self.recurrent:
            self.train_prev_recurrent_states = None
            self.train_recurrent_states = self._restart_ended(
                self.train_recurrent_states, batch_done, batch_reset)

    def _per_lookahead_ok(self):
        rbuf, up = self.replay_buffer, self.replay_updater
        return (os.environ.get("PFRL_PER_LOOKAHEAD", "1") != "0" and self._replay_stream is not None
                and self.use_graphs and not self.recurrent and hasattr(rbuf, "sample_prepare")
                # (somebody who wrapped sample() to see the minibatches must keep seeing them)
                and "sample" not in vars(rbuf)
                and getattr(rbuf, "store", None) is not None and not up.episodic_update
                and up.n_times_update == 1 and up.update_func == self.update)

    def _batch_observe_train_per(self, batch_obs, batch_reward, batch_done, batch_reset):
        """The loop of _batch_observe_train (reference :516-549) for a prioritized device buffer,
        with the host one update point ahead of the device.  What the next forward pass waits for
        is a chain on the replay stream -- priorities of minibatch k, pending appends, B dependent
        draws, gather -- and only its first link needs minibatch k's TD errors.  Launched in
        program order, that chain started 80 us after the forward graph had ended: the host was
        still walking through the appends and the sample's preparation (tools/pipeline_events.py).
        Here, once sample k is launched, the host first appends the transitions up to update
        point k + 1 and prepares that sample (NumPy draws, staging transfer, table rows: nothing
        the update changes), THEN launches update k; what is left for afterwards is two launches.
        Appends, draws from the NumPy stream, target syncs and updates keep the reference's order;
        the look-ahead stops where that could not be guaranteed (a target sync or a tree frame
        change ahead: both must see update k done)."""
        rbuf, up = self.replay_buffer, self.replay_updater
        tree = rbuf.memory.tree
        n = len(batch_obs)
        B = up.batchsize
        ahead = int(getattr(rbuf, "num_steps", 1)) + 1      # entries one transition can append

        def due():      # ReplayUpdater.update_if_necessary
            return len(rbuf) >= up.replay_start_size and self.t % up.update_interval == 0

        def advance(i):
            self.t += 1
            self._cumulative_steps += 1
            if self.t % self.target_update_interval == 0:
                self.sync_target_network()
            self._append_transition(i, batch_obs, batch_reward, batch_done, batch_reset)

        i = 0
        finish_next = None          # the prepared sample of the update point the loop stands on
        while True:
            if finish_next is None:
                if i >= n:
                    break
                advance(i)
                i += 1
                if not due():
                    continue
                finish_next = rbuf.sample_prepare(B)
            experiences = finish_next()
            finish_next = None
            # the gather goes onto the replay stream right behind the draws, ahead of the rows
            # the look-ahead is about to ship
            gathered = batch_experiences(experiences, device=self.device, phi=self.phi,
                                         gamma=self.gamma, batch_states=self.batch_states)
            while (i < n and (self.t + 1) % self.target_update_interval != 0
                   and tree.next_appends_keep_frame(ahead)):
                advance(i)
                i += 1
                if due():
                    finish_next = rbuf.sample_prepare(B)
                    break
            self.update(experiences, _gathered=gathered)

    def _batch_observe_train_fused(self, batch_obs, batch_reward, batch_done, batch_reset):
        """Same schedule as the loop above (reference :516-549), reorganised
        for the device: pass 1 appends all N transitions and draws the index
        sets of every update this step is going to make (uniform replay: they
        depend only on len(buffer) and the NumPy stream, in the same order);
        ONE fused launch gathers all minibatches; pass 2 walks the envs again
        and runs target syncs and updates at exactly their original positions.
        The entry ring keeps `slack` spare slots, so rows appended "early" never
        overwrite entries still visible to an earlier update of this step."""
        rbuf = self.replay_buffer
        n_env = len(batch_obs)
        if self.device_step:
            from pfrl_amd.agents import _dqn_device_step

            _dqn_device_step.begin_observe(self, batch_obs, batch_reward, batch_done, batch_reset)
        # The step is processed in a few env ranges, a small one first: while the GPU
        # runs the first range's updates the host prepares the next range (appends,
        # index draws, launches), instead of the GPU idling through the whole
        # preparation.  Order of appends, RNG draws, target syncs and updates is the
        # reference's in every case.
        chunks = self.step_fused_chunks
        side = None
        if self.__dict__.get("_obs_cols") is not None and not self._chunks_set_by_caller:
            # native step: the host's share of a step is ~0.3 ms and nothing waits for the GPU, so
            # there is no preparation to hide -- one range, one gather, one graph per step
            # (measured 33.4 k vs 33.0 k env-steps/s with the (0.1, 0.4) cuts)
            chunks = ()
            if self._range_overlap_ok():
                # (opt-in, PFRL_DQN_RANGE_OVERLAP=1: the NEXT range's replay side -- appends, the
                # fused gather, the target network's pass over its next-states: 0.5 ms of a 6.1 ms
                # step -- on a second stream UNDER the current range's update graph.  Measured on
                # MI355X and NOT faster: 37.9 k vs 41.3 k env-steps/s at cuts of 12.5 / 25 / 50 %.
                # The 64 B = 32 updates are latency-bound launch chains, and a 38-GFLOP target pass
                # beside them stretches every launch (update 84.5 -> 92.4 us) by more than the
                # 0.4 ms it hides.)
                chunks = self._RANGE_OVERLAP_CUT
                side = self._range_side_stream()
        cuts = sorted({0, n_env} | {int(n_env * f) for f in chunks})
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            self._observe_range_fused(lo, hi, batch_obs, batch_reward, batch_done, batch_reset,
                                      side=side)

    _RANGE_OVERLAP_CUT = (float(os.environ.get("PFRL_DQN_RANGE_CUT", "0.25")),)

    def _range_overlap_ok(self):
        return (os.environ.get("PFRL_DQN_RANGE_OVERLAP", "0") == "1" and self.device.type == "cuda"
                and self.use_graphs and self.range_graphs and self._replay_stream is None
                and self.batch_target_pass and not self.recurrent)

    def _range_side_stream(self):
        s = self.__dict__.get("_range_side")
        if s is None:
            s = self._range_side = torch.cuda.Stream(self.device)
        return s

    def _observe_range_fused(self, lo, hi, batch_obs, batch_reward, batch_done, batch_reset,
                             side=None):
        """``side``: a second stream for this range's replay side (appends, gather, target pass).
        It waits for the main stream at the first range of a step (this step's frames, actions and
        everything before them) and after a target sync; the main stream waits for it before the
        range's updates.  Nothing the side stream writes is read by the update graph of the
        PREVIOUS range still running on the main stream (minibatch and target buffers alternate
        between sets, the table rows it appends are read by gathers only), nothing it reads is
        written there (the target network, the frame ring)."""
        import contextlib

        main = torch.cuda.current_stream(self.device) if side is not None else None
        if side is not None and (lo == 0 or self._side_needs_main):
            side.wait_stream(main)
            self._side_needs_main = False
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            plan = self._plan_and_fetch_range(lo, hi, batch_obs, batch_reward, batch_done, batch_reset)
        if side is not None:
            main.wait_stream(side)
        self._run_range_updates(lo, hi, plan)

    def _plan_and_fetch_range(self, lo, hi, batch_obs, batch_reward, batch_done, batch_reset):
        rbuf = self.replay_buffer
        up = self.replay_updater
        t0 = self.t
        plan_env, plan_seqs = [], []
        native = None
        if self.device_step and self.__dict__.get("_obs_cols") is not None:
            from pfrl_amd.agents import _dqn_device_step

            native = _dqn_device_step.plan_range(self, lo, hi, batch_obs)
        if native is not None:
            # appends, queue bookkeeping and every index set of the range came from ONE native
            # call and ONE transfer (same NumPy stream use as the loops below)
            plan_env, slots_dev, U = native
        elif self._batched_append_ok(lo, hi, batch_obs):
            # the per-env loop below as array writes: every env of the range has a pending
            # transition, entries are one transition long, so len(buffer) and the queue head at
            # each point of the loop are known in advance and the index draws (same NumPy
            # stream, same order) can follow the appends
            prev = self._last_obs_batch
            lens, heads = rbuf.append_batch_n1(
                prev.refs[lo:hi], prev.min_seq[lo:hi],
                np.asarray(self.batch_last_action[lo:hi], dtype=np.int64),
                np.asarray(batch_reward[lo:hi], dtype=np.float64),
                batch_obs.refs[lo:hi], batch_obs.min_seq[lo:hi],
                np.asarray(batch_done[lo:hi], dtype=np.uint8))
            for i in range(lo, hi):
                if batch_reset[i] or batch_done[i]:
                    self.batch_last_obs[i] = None
                    self.batch_last_action[i] = None
                j = i - lo
                if (lens[j] >= up.replay_start_size
                        and (t0 + j + 1) % up.update_interval == 0):
                    for _ in range(up.n_times_update):
                        plan_env.append(i)
                        plan_seqs.append(rbuf.lookahead_sample_at(lens[j], heads[j], up.batchsize))
        else:
            for i in range(lo, hi):
                self._append_transition(i, batch_obs, batch_reward, batch_done, batch_reset)
                if (len(rbuf) >= up.replay_start_size
                        and (t0 + (i - lo) + 1) % up.update_interval == 0):
                    for _ in range(up.n_times_update):
                        plan_env.append(i)
                        plan_seqs.append(rbuf.lookahead_sample(up.batchsize))
        if native is not None:
            big = rbuf.store.fetch_many_slots(slots_dev, U, up.batchsize, self.phi,
                                              self.gamma) if U else None
        else:
            big = rbuf.fetch_many(plan_seqs, self.phi, self.gamma) if plan_seqs else None
        if big is not None and self.batch_target_pass and self._target_is_deterministic():
            # no target sync may fall inside this range (the targets would go stale)
            tui = self.target_update_interval
            if (t0 + (hi - lo)) // tui == t0 // tui:
                ns = big["next_state"]
                U, B = ns.shape[0], ns.shape[1]
                raw = self._precompute_target_raw(ns.view((U * B,) + tuple(ns.shape[2:])))
                # captured updates are keyed by buffer addresses: park the values in a
                # persistent buffer instead of whatever block the allocator handed out
                key = (U, B, getattr(rbuf.store, "many_parity", 0)) + tuple(raw.shape[1:])
                buf = self._target_raw_bufs.get(key)
                if buf is None:
                    buf = self._target_raw_bufs[key] = torch.empty(
                        (U, B) + tuple(raw.shape[1:]), dtype=raw.dtype, device=raw.device)
                buf.view(raw.shape).copy_(raw)
                big["target_next_raw"] = buf
        return plan_env, big, t0

    def _run_range_updates(self, lo, hi, plan):
        plan_env, big, t0 = plan
        if big is not None and self._range_as_one_graph(t0, hi - lo):
            # nothing happens on the host between this range's updates (no target sync
            # inside it): the whole range replays as ONE captured graph
            try:
                losses, ys = self._graphed.run_range(big)
            except Exception as e:
                # a model / optimizer that cannot be captured: as in _graphed_step, stay on the
                # GPU without the range graph (the per-update path below has its own eager
                # fallback).  The capture restored its snapshot; no counter has moved yet.
                if self._graphed.graphs:
                    raise
                self.logger.warning("HIP-graph capture of an env range failed (%s); updating "
                                    "one minibatch at a time", e)
                self.range_graphs = False
            else:
                self.t += hi - lo
                self._cumulative_steps += hi - lo
                # graph-owned outputs: the next replay overwrites them
                self.loss_record.extend(losses.clone())
                self.q_record.extend(ys.clone())
                self.optim_t += len(plan_env)
                return
        p = 0
        deferred = [] if self.use_graphs else None
        # A range that normally replays as one graph but has a target sync inside it this
        # time (once per target_update_interval) runs update by update.  Capturing one graph
        # per minibatch slot for that rare case would cost tens of captures the first time
        # each slot is met: instead every minibatch is copied into ONE persistent staging
        # set (7 MB, a few us) and a single captured update is replayed on it.
        staged = (big is not None and self.use_graphs and self.range_graphs
                  and self._graphed is not None and self._graphed.range_capturable())
        if staged:
            deferred = None
        for i in range(lo, hi):
            self.t += 1
            self._cumulative_steps += 1
            if self.t % self.target_update_interval == 0:
                self.sync_target_network()
            while p < len(plan_env) and plan_env[p] == i:
                mb = {k: v[p] for k, v in big.items()}
                if staged:
                    mb = self._stage_minibatch(mb)
                self._update_from_batch(mb, deferred=deferred)
                p += 1
        if deferred and self.use_graphs:
            # every update above replayed its own graph, so all outputs are still live
            if len({id(l) for l, _ in deferred}) == len(deferred):
                self.loss_record.extend(torch.stack([l.reshape(()) for l, _ in deferred]))
                self.q_record.extend(torch.cat([y.reshape(-1) for _, y in deferred]))
            else:   # graphs were shared (should not happen): last values only
                for l, y in deferred:
                    self.loss_record.extend(l.clone())
                    self.q_record.extend(y.clone())

    def _stage_minibatch(self, mb):
        """Copy one minibatch (dict of tensors) into the persistent staging set."""
        key = tuple(sorted((k, tuple(v.shape), v.dtype, v.stride()) for k, v in mb.items()))
        bufs = self._single_bufs.get(key)
        if bufs is None:
            bufs = self._single_bufs[key] = {k: torch.empty_strided(v.shape, v.stride(), dtype=v.dtype,
                                                                    device=v.device)
                                             for k, v in mb.items()}
        for k, v in mb.items():
            bufs[k].copy_(v)
        return bufs

    def _batched_append_ok(self, lo, hi, batch_obs):
        from pfrl_amd.device_store import DeviceObsBatch

        prev = getattr(self, "_last_obs_batch", None)
        rbuf = self.replay_buffer
        return (not self.recurrent and isinstance(batch_obs, DeviceObsBatch)
                and isinstance(prev, DeviceObsBatch) and prev.store is batch_obs.store
                and len(prev) == len(batch_obs)
                and hasattr(rbuf, "batch_append_supported") and rbuf.batch_append_supported(batch_obs)
                and all(o is not None for o in self.batch_last_obs[lo:hi]))

    def _range_as_one_graph(self, t0, n):
        """All updates of an env range as one HIP graph: graphs on, no prioritized replay
        (priorities feed back between updates), no data-parallel collective between
        backward and step, and no target sync inside the range."""
        if not (self.use_graphs and self.range_graphs):
            return False
        tui = self.target_update_interval
        if (t0 + n) // tui != t0 // tui:
            return False
        if self._graphed is None:
            from pfrl_amd.agents.graphed_update import GraphedUpdate

            self._graphed = GraphedUpdate(self)
            self._graphed.pipeline = self._replay_stream is not None
        return self._graphed.range_capturable()

    def _batch_observe_eval(self, batch_obs, batch_reward, batch_done, batch_reset):
        if self.recurrent:
            self.test_recurrent_states = self._restart_ended(
                self.test_recurrent_states, batch_done, batch_reset)

    def batch_observe(self, batch_obs, batch_reward, batch_done, batch_reset):
        if self.training:
            self._in_train_step = True
            try:
                return self._batch_observe_train(batch_obs, batch_reward, batch_done, batch_reset)
            finally:
                self._in_train_step = False
                self._flush_backward()   # nothing of a step's updates is pending outside of it
        return self._batch_observe_eval(batch_obs, batch_reward, batch_done, batch_reset)

    def _can_start_replay(self):
        if len(self.replay_buffer) < self.replay_start_size:
            return False
        return not self.recurrent or self.replay_buffer.n_episodes >= self.minibatch_size

    def stop_episode(self):
        if self.recurrent:
            self.test_recurrent_states = None

    # -- persistence / statistics ----------------------------------------------
    def save_snapshot(self, dirname):
        self.save(dirname)
        torch.save(self.t, os.path.join(dirname, "t.pt"))
        torch.save(self.optim_t, os.path.join(dirname, "optim_t.pt"))
        torch.save(self._cumulative_steps, os.path.join(dirname, "_cumulative_steps.pt"))
        self.replay_buffer.save(os.path.join(dirname, "replay_buffer.pkl"))

    def load_snapshot(self, dirname):
        self.load(dirname)
        self.t = torch.load(os.path.join(dirname, "t.pt"))
        self.optim_t = torch.load(os.path.join(dirname, "optim_t.pt"))
        self._cumulative_steps = torch.load(os.path.join(dirname, "_cumulative_steps.pt"))
        self.replay_buffer.load(os.path.join(dirname, "replay_buffer.pkl"))

    def get_statistics(self):
        return [
            ("average_q", _mean_or_nan(self.q_record.values())),
            ("average_loss", _mean_or_nan(self.loss_record.values())),
            ("cumulative_steps", self.cumulative_steps),
            ("n_updates", self.optim_t),
            ("rlen", len(self.replay_buffer)),
        ]
