"""Implicit Quantile Networks (https://arxiv.org/abs/1806.06923) on the DQN device path.

Mirrors ``pfrl.agents.iqn`` (/root/reference/pfrl/agents/iqn.py): cosine embedding
(:11-60), ``ImplicitQuantileQFunction`` (:63-124), the quantile Huber loss (:176-206)
and its accumulation (:211-255), ``IQN`` (:258-433).  Replay, sampling, gathers and the
captured update are DQN's; the quantile thresholds are drawn with ``torch.rand`` on
the agent's device, three draws per update in the reference's order (taus for the
prediction, taus~ for the greedy action, taus' for the target).
"""
import math

import torch
from torch import nn

from pfrl_amd.action_value import QuantileDiscreteActionValue
from pfrl_amd.agents import dqn
from pfrl_amd.nn.recurrent import Recurrent


def cosine_basis_functions(x, n_basis_functions=64):
    """cos(pi * i * x) for i = 1..n (the paper's eq. 4 with the corrected index range)."""
    i_pi = torch.arange(1, n_basis_functions + 1, dtype=torch.float, device=x.device) * math.pi
    return torch.cos(x[..., None] * i_pi)


class CosineBasisLinear(nn.Module):
    """Linear layer on the cosine embedding of quantile thresholds: (...,) -> (..., out)."""

    def __init__(self, n_basis_functions, out_size):
        super().__init__()
        self.linear = nn.Linear(n_basis_functions, out_size)
        self.n_basis_functions = n_basis_functions
        self.out_size = out_size

    def forward(self, x):
        emb = cosine_basis_functions(x, self.n_basis_functions)
        out = self.linear(emb.reshape(-1, self.n_basis_functions))
        return out.reshape(*x.shape, self.out_size)


class ImplicitQuantileQFunction(nn.Module):
    """``psi`` embeds the state, ``phi`` the thresholds, ``f`` maps their product to
    per-action quantiles.  Calling the module returns a function of the thresholds."""

    def __init__(self, psi, phi, f):
        super().__init__()
        self.psi, self.phi, self.f = psi, phi, f

    def forward(self, x):
        psi_x = self.psi(x)
        assert psi_x.ndim == 2 and psi_x.shape[0] == x.shape[0]
        return _quantile_head(psi_x, self.phi, self.f)


def _quantile_head(psi_x, phi, f):
    """The function of the thresholds both Q-function classes return: quantile values of every
    action at ``taus`` (B, n_taus) for the embedded states ``psi_x`` (B, hidden)."""

    def evaluate_with_quantile_thresholds(taus):
        batch, hidden = psi_x.shape
        assert taus.ndim == 2 and taus.shape[0] == batch
        phi_taus = phi(taus)
        assert phi_taus.shape == (batch, taus.shape[1], hidden)
        h = f((psi_x.unsqueeze(1) * phi_taus).reshape(-1, hidden))
        return QuantileDiscreteActionValue(h.reshape(batch, taus.shape[1], h.shape[-1]))

    return evaluate_with_quantile_thresholds


class RecurrentImplicitQuantileQFunction(Recurrent, nn.Module):
    """``ImplicitQuantileQFunction`` whose state embedding ``psi`` is a recurrent module
    (``RecurrentSequential`` ...): ``forward(packed_obs, recurrent_state)`` returns the threshold
    function over the flat, time-major batch of all steps, and the new state
    (reference :127-173)."""

    def __init__(self, psi, phi, f):
        super().__init__()
        self.psi, self.phi, self.f = psi, phi, f

    def forward(self, x, recurrent_state):
        packed, recurrent_state = self.psi(x, recurrent_state)
        assert isinstance(packed, nn.utils.rnn.PackedSequence)
        psi_x = packed.data
        assert psi_x.ndim == 2
        return _quantile_head(psi_x, self.phi, self.f), recurrent_state


def compute_eltwise_huber_quantile_loss(y, t, taus):
    """|tau - 1[t < y]| * huber(y, t) for every (prediction, target) pair:
    y (B, N), t (B, N'), taus (B, N) -> (B, N, N')."""
    assert y.shape == taus.shape
    y, t, taus = torch.broadcast_tensors(y.unsqueeze(2), t.unsqueeze(1), taus.unsqueeze(2))
    below = (t < y).float()
    return torch.abs(taus - below) * nn.functional.smooth_l1_loss(y, t, reduction="none")


def compute_value_loss(eltwise_loss, batch_accumulator="mean"):
    assert batch_accumulator in ("mean", "sum") and eltwise_loss.ndim == 3
    if batch_accumulator == "sum":
        return eltwise_loss.mean(2).sum()
    return eltwise_loss.mean((0, 2)).sum()


def compute_weighted_value_loss(eltwise_loss, weights, batch_accumulator="mean"):
    assert batch_accumulator in ("mean", "sum") and eltwise_loss.ndim == 3
    loss_sum = torch.matmul(eltwise_loss.mean(2).sum(1), weights)
    return loss_sum / eltwise_loss.shape[0] if batch_accumulator == "mean" else loss_sum


class IQN(dqn.DQN):
    """DQN arguments plus ``quantile_thresholds_N`` (64), ``quantile_thresholds_N_prime``
    (64), ``quantile_thresholds_K`` (32) and ``act_deterministically`` (False)."""

    _fused_td_double = None   # quantile regression: not the scalar TD loss

    def __init__(self, *args, **kwargs):
        self.quantile_thresholds_N = kwargs.pop("quantile_thresholds_N", 64)
        self.quantile_thresholds_N_prime = kwargs.pop("quantile_thresholds_N_prime", 64)
        self.quantile_thresholds_K = kwargs.pop("quantile_thresholds_K", 32)
        self.act_deterministically = kwargs.pop("act_deterministically", False)
        # the target network is a function of freshly drawn thresholds: no step batching
        kwargs.setdefault("batch_target_pass", False)
        super().__init__(*args, **kwargs)

    def _rand(self, rows, cols):
        return torch.rand(rows, cols, device=self.device, dtype=torch.float)

    def _compute_target_values(self, exp_batch):
        batch_size = exp_batch["reward"].shape[0]
        taus_tilde = self._rand(batch_size, self.quantile_thresholds_K)
        target_next_tau2av = self._action_value(self.target_model, exp_batch["next_state"],
                                                exp_batch.get("next_recurrent_state"))
        greedy_actions = target_next_tau2av(taus_tilde).greedy_actions
        taus_prime = self._rand(batch_size, self.quantile_thresholds_N_prime)
        next_maxz = target_next_tau2av(taus_prime).evaluate_actions_as_quantiles(greedy_actions)
        return (exp_batch["reward"].unsqueeze(-1) + exp_batch["discount"].unsqueeze(-1)
                * (1.0 - exp_batch["is_state_terminal"].unsqueeze(-1)) * next_maxz)

    def _compute_y_and_taus(self, exp_batch):
        tau2av = self._action_value(self.model, exp_batch["state"],
                                    exp_batch.get("recurrent_state"))
        taus = self._rand(exp_batch["reward"].shape[0], self.quantile_thresholds_N)
        av = tau2av(taus)
        self._q_all = av.q_values.detach()
        return av.evaluate_actions_as_quantiles(exp_batch["action"]), taus

    def _compute_loss(self, exp_batch, errors_out=None, want_errors=False, record=True):
        y, taus = self._compute_y_and_taus(exp_batch)
        self._last_y = self._q_all.reshape(-1)
        if record:
            self.q_record.extend(self._last_y)
        with torch.no_grad():
            t = self._compute_target_values(exp_batch)
        eltwise_loss = compute_eltwise_huber_quantile_loss(y, t, taus)
        delta = None
        if errors_out is not None or want_errors:
            delta = eltwise_loss.detach().mean((1, 2))
            if errors_out is not None:
                del errors_out[:]
                errors_out.extend(delta.cpu().numpy())
        if "weights" in exp_batch:
            loss = compute_weighted_value_loss(eltwise_loss, exp_batch["weights"],
                                               batch_accumulator=self.batch_accumulator)
        else:
            loss = compute_value_loss(eltwise_loss, batch_accumulator=self.batch_accumulator)
        return loss, delta

    def _evaluate_model(self, batch_obs):
        tau2av = super()._evaluate_model(batch_obs)     # one step; recurrent states handled there
        n = len(batch_obs)
        if not self.training and self.act_deterministically:
            taus_tilde = torch.linspace(start=0, end=1, steps=self.quantile_thresholds_K,
                                        device=self.device, dtype=torch.float).repeat(n, 1)
        else:
            taus_tilde = self._rand(n, self.quantile_thresholds_K)
        return tau2av(taus_tilde)
