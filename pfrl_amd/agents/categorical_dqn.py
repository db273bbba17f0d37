"""Categorical (C51) DQN and its Double variant -- the Rainbow update.

Mirrors ``pfrl.agents.categorical_dqn`` (/root/reference/pfrl/agents/
categorical_dqn.py: projection :7-57, losses :60-104, agent :107-204) and
``categorical_double_dqn.py`` (:7-52).  On the GPU the categorical projection, the
cross-entropy loss, its gradient and the per-sample KL priorities are one HIP
launch (``pfrl_c51_loss``); the stock-PyTorch composite below is the CPU path and
the path of subclasses that override the target computation.  Both plug into
the same device replay path as DQN (fused gather, HIP-graph update, PER
priorities handed over as a device tensor).
"""
import os
from contextlib import nullcontext as _nullcontext

import torch

from pfrl_amd.agents import dqn
from pfrl_amd.utils.contexts import evaluating


def _apply_categorical_projection(y, y_probs, z):
    """Algorithm 1 of https://arxiv.org/abs/1707.06887: project the atoms
    ``y`` (batch, n_atoms) carrying mass ``y_probs`` onto the fixed, evenly
    spaced support ``z`` (n_atoms,)."""
    batch_size, n_atoms = y.shape
    assert z.shape == (n_atoms,)
    assert y_probs.shape == (batch_size, n_atoms)
    delta_z = z[1] - z[0]
    v_min, v_max = z[0], z[-1]
    y = torch.clamp(y, v_min, v_max)
    bj = torch.clamp((y - v_min) / delta_z, 0, n_atoms - 1)   # guards inexact delta_z
    lo, up = torch.floor(bj), torch.ceil(bj)
    z_probs = torch.zeros((batch_size, n_atoms), dtype=torch.float32, device=y.device)
    offset = torch.arange(0, batch_size * n_atoms, n_atoms, dtype=torch.int32,
                          device=y.device)[..., None]
    frac = bj - lo
    # mass to the lower neighbour uses 1 - (bj - l) so that integer bj keeps all its mass
    z_probs.view(-1).scatter_add_(0, (lo.long() + offset).view(-1), (y_probs * (1 - frac)).view(-1))
    z_probs.view(-1).scatter_add_(0, (up.long() + offset).view(-1), (y_probs * frac).view(-1))
    return z_probs


def compute_value_loss(eltwise_loss, batch_accumulator="mean"):
    assert batch_accumulator in ("mean", "sum")
    if batch_accumulator == "sum":
        return eltwise_loss.sum()
    return eltwise_loss.sum(dim=1).mean()


def compute_weighted_value_loss(eltwise_loss, batch_size, weights, batch_accumulator="mean"):
    assert batch_accumulator in ("mean", "sum")
    loss_sum = torch.matmul(eltwise_loss.sum(dim=1), weights.to(eltwise_loss.device))
    if batch_accumulator == "mean":
        return loss_sum / batch_size
    return loss_sum


class CategoricalDQN(dqn.DQN):
    """q_function must return DistributionalDiscreteActionValue; clip_delta is
    ignored (reference :107-113)."""

    _fused_td_double = None   # cross-entropy on distributions: not the scalar TD loss
    _c51_double = False       # greedy next action: target net (False) / online net (True)
    _side_passes = None

    def _pass_streams(self):
        """Two side streams for the no-grad passes of the Double update (None: switched off
        with PFRL_C51_FORK=0, or a precomputed target pass is in the minibatch)."""
        if os.environ.get("PFRL_C51_FORK", "1") == "0" or self.device.type != "cuda":
            return None
        if self._side_passes is None:
            self._side_passes = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
        return self._side_passes

    def _project(self, exp_batch, next_dist, z_values):
        Tz = (exp_batch["reward"][..., None]
              + (1.0 - exp_batch["is_state_terminal"][..., None])
              * torch.unsqueeze(exp_batch["discount"], 1) * z_values[None])
        return _apply_categorical_projection(Tz, next_dist, z_values)

    def _compute_target_values(self, exp_batch):
        target_next_qout = self._target_next_action_value(exp_batch)
        next_q_max = target_next_qout.max_as_distribution.detach()
        return self._project(exp_batch, next_q_max, target_next_qout.z_values)

    def _compute_y_and_t(self, exp_batch):
        qout = self._action_value(self.model, exp_batch["state"], exp_batch.get("recurrent_state"))
        batch_actions = exp_batch["action"]
        batch_q = qout.evaluate_actions_as_distribution(batch_actions)
        with torch.no_grad():
            batch_q_target = self._compute_target_values(exp_batch)
            self._q_scalars = qout.evaluate_actions(batch_actions).detach()
        return batch_q, batch_q_target

    def _fused_c51_applicable(self):
        cls = type(self)
        stock = (CategoricalDQN, CategoricalDoubleDQN)
        return (self.fused_td_loss and self.device.type == "cuda"
                and any(cls._compute_target_values is c._compute_target_values for c in stock)
                and cls._compute_y_and_t is CategoricalDQN._compute_y_and_t
                and cls._project is CategoricalDQN._project)

    def _compute_loss_fused_c51(self, exp_batch, errors_out, record):
        """Same quantities as the composite path below out of ONE launch
        (pfrl_c51_loss): projection, cross entropy, its gradient, Q(s, a), KL."""
        from pfrl_amd import ops

        # The three network passes of the Double update (online on s; target and online on s')
        # read nothing of each other: the two no-grad passes are issued on side streams forked
        # from the current one and joined in front of the loss launch.  In a captured update they
        # become parallel branches of the graph; the host order of the calls -- hence the order in
        # which the NoisyNet layers consume the device generator -- is that of the reference
        # (pfrl/agents/categorical_double_dqn.py:17-39 after categorical_dqn.py:165-170).
        cur = torch.cuda.current_stream(self.device)
        side = self._pass_streams() if type(self)._c51_double else None
        if side is not None:
            for st in side:
                st.wait_stream(cur)
        qout = self.model(exp_batch["state"])
        if not ops.c51_loss_supported(qout.q_dist):
            if side is not None:
                for st in side:
                    cur.wait_stream(st)
            return None
        with torch.no_grad():
            if type(self)._c51_double:
                with evaluating(self.target_model), evaluating(self.model):
                    with torch.cuda.stream(side[0]) if side is not None else _nullcontext():
                        target_next = self._target_next_action_value(exp_batch)
                    with torch.cuda.stream(side[1]) if side is not None else _nullcontext():
                        select = self.model(exp_batch["next_state"]).q_dist
                if side is not None:
                    for st in side:
                        cur.wait_stream(st)
                    # allocated on the side streams, read by the loss launch on this one
                    target_next.q_dist.record_stream(cur)
                    select.record_stream(cur)
            else:
                target_next = self._target_next_action_value(exp_batch)
                select = None
        loss, qsa, delta = ops.c51_loss(
            qout.q_dist, exp_batch["action"], target_next.q_dist, select, target_next.z_values,
            exp_batch["reward"], exp_batch["discount"], exp_batch["is_state_terminal"],
            exp_batch.get("weights"), self.batch_accumulator == "mean")
        self._analytic_backward = (qout.q_dist, loss.grad_fn.saved_tensors[0]
                                   if loss.grad_fn is not None else None)
        self._q_scalars = qsa
        self._last_y = qsa
        if record:
            self.q_record.extend(qsa)
        if errors_out is not None:
            del errors_out[:]
            errors_out.extend(delta.cpu().numpy())
        return loss, delta

    def _compute_loss(self, exp_batch, errors_out=None, want_errors=False, record=True):
        if self._fused_c51_applicable():
            out = self._compute_loss_fused_c51(exp_batch, errors_out, record)
            if out is not None:
                return out
        y, t = self._compute_y_and_t(exp_batch)
        self._last_y = self._q_scalars
        if record:
            self.q_record.extend(self._q_scalars)
        # cross entropy; y clipped to avoid log(0)
        eltwise_loss = -t * torch.log(torch.clamp(y, 1e-10, 1.0))
        delta = None
        if errors_out is not None or want_errors:
            delta = eltwise_loss.detach().sum(dim=1)   # prioritise by KL divergence
            if errors_out is not None:
                del errors_out[:]
                errors_out.extend(delta.cpu().numpy())
        if "weights" in exp_batch:
            loss = compute_weighted_value_loss(eltwise_loss, y.shape[0], exp_batch["weights"],
                                               batch_accumulator=self.batch_accumulator)
        else:
            loss = compute_value_loss(eltwise_loss, batch_accumulator=self.batch_accumulator)
        return loss, delta


class CategoricalDoubleDQN(CategoricalDQN):
    """Action chosen by the online network, distribution taken from the target
    network (reference categorical_double_dqn.py:10-52)."""

    _c51_double = True

    def _compute_target_values(self, exp_batch):
        batch_next_state = exp_batch["next_state"]
        with evaluating(self.target_model), evaluating(self.model):
            target_next_qout = self._target_next_action_value(exp_batch)
            next_qout = self._action_value(self.model, batch_next_state,
                                           exp_batch.get("next_recurrent_state"))
        next_q_max = target_next_qout.evaluate_actions_as_distribution(
            next_qout.greedy_actions.detach())
        return self._project(exp_batch, next_q_max, target_next_qout.z_values)
