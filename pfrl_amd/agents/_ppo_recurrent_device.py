"""Recurrent PPO on the device rollout (reference pfrl/agents/ppo.py:56-107,196-225,534-632).

The host path (``agents/ppo_host.py``) restates the reference's list-of-dicts algorithm: every
transition a dict, every fragment a list, sequences re-collated per minibatch.  Here the rollout
stays what it is for the feed-forward agent -- ``[T][N]`` columns, observations as frame-ring
slots in HBM (``agents/ppo.py::_Rollout``) -- plus two columns of recurrent states (the state
each step started from and the state it produced, reference :777-784), and a "sequence" is an
array of flat positions ``t * N + env``:

* fragments (done / reset / rollout end, reference :786-789, :448-456) come from the rollout's
  cut bookkeeping in the reference's ``memory`` order;
* the value pass packs ALL fragments longest-first exactly as ``pack_sequence`` would lay them
  out (time-major positions + ``batch_sizes`` computed on the host, ONE gather of the observations
  on the device, the stored start states picked by index) and scatters log pi(a|s), V(s) and
  V(s') back into the columns;
* GAE is ``pfrl_gae_scan`` mode 2: the reference stores ``float(v)`` for a recurrent dataset
  (:98-107), so :36-47 runs in f64 throughout (tests/golden/gae_recurrent.npz);
* minibatches follow the reference's walk -- ``_limit_sequence_length``, ``random.shuffle`` per
  epoch on the module-level generator, ``_yield_subset_of_sequences_with_fixed_number_of_items``
  -- on the position arrays, so the same sequences meet in the same minibatches.

Nothing of a rollout is looked at on the host except integer bookkeeping.
"""
import random

import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence

from pfrl_amd import ops
from pfrl_amd.agents.ppo_host import (_limit_sequence_length,
                                      _yield_subset_of_sequences_with_fixed_number_of_items)
from pfrl_amd.utils.clip_l2_grad_norm import clip_grad_norm_device_
from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.recurrent import (_map_state, mask_recurrent_state_at, one_step_forward,
                                      unwrap_packed_sequences_recursive)


def _leaves(state, out=None):
    """Tensors of a (nested tuple) recurrent state, depth first."""
    out = [] if out is None else out
    if state is None:
        return out
    if isinstance(state, torch.Tensor):
        out.append(state)
    else:
        for s in state:
            _leaves(s, out)
    return out


def _rebuild(template, leaves):
    """``leaves`` (an iterator) arranged like ``template``."""
    if isinstance(template, torch.Tensor):
        return next(leaves)
    return tuple(_rebuild(s, leaves) for s in template)


class StateColumns:
    """Recurrent states of every rollout step: per leaf tensor one buffer ``[T][layers][N][H]``
    (zeros where the reference holds ``None``: ``concatenate_recurrent_states`` fills zeros there)."""

    def __init__(self, t_cap):
        self.t_cap = t_cap
        self.template = None
        self.bufs = None

    def write(self, t, state):
        if state is None:
            if self.bufs is not None:
                for b in self.bufs:
                    b[t].zero_()
            return
        if self.bufs is None:
            self.template = _map_state(lambda s: s, state)
            self.bufs = [torch.zeros((self.t_cap,) + tuple(leaf.shape), dtype=leaf.dtype,
                                     device=leaf.device) for leaf in _leaves(state)]
        for b, leaf in zip(self.bufs, _leaves(state)):
            b[t].copy_(leaf.detach())

    def pick(self, t_idx, e_idx):
        """The batched state of sequences starting at (t_idx[i], e_idx[i]); None if no state was
        ever stored."""
        if self.bufs is None:
            return None
        picked = [b[t_idx, :, e_idx].transpose(0, 1).contiguous() for b in self.bufs]
        return _rebuild(self.template, iter(picked))


def packed_layout(seqs):
    """``seqs``: list of int64 position arrays.  Returns (order of the sequences longest-first and
    stable, as ``sorted(key=len, reverse=True)``; positions in ``pack_sequence``'s time-major
    order; batch_sizes) -- reference ``flatten_sequences_time_first`` + ``pack_sequence``."""
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    order = np.argsort(-lens, kind="stable")
    slen = lens[order]
    maxlen = int(slen[0])
    # counts[t] = sequences longer than t (a prefix of the sorted list)
    counts = np.searchsorted(-slen, -np.arange(1, maxlen + 1), side="right")
    pad = np.full((len(seqs), maxlen), -1, dtype=np.int64)
    for row, i in enumerate(order):
        pad[row, :lens[i]] = seqs[i]
    flat = pad.T.reshape(-1)
    return order, flat[flat >= 0], counts.astype(np.int64)


class RecurrentDeviceRollouts:
    """Acting / observing / updating of ``PPO(recurrent=True, gpu >= 0)`` on the device rollout."""

    def __init__(self, agent):
        self.agent = agent
        self.train_state = None          # state after the last step (ended envs zeroed)
        self.train_prev_state = None
        self.test_state = None
        self.prev_cols = None
        self.next_cols = None

    # -- acting ------------------------------------------------------------------------------
    def act_train(self, refs_dev):
        a = self.agent
        x = a._features(refs_dev)
        with torch.no_grad(), evaluating(a.model):
            assert self.train_prev_state is None
            self.train_prev_state = self.train_state
            (distrib, value), self.train_state = one_step_forward(a.model, x, self.train_prev_state)
            action = a._sample_action(distrib)
            a.entropy_record.extend(distrib.entropy())
            a.value_record.extend(value)
        return action

    def act_eval(self, x):
        a = self.agent
        with torch.no_grad(), evaluating(a.model):
            (distrib, _), self.test_state = one_step_forward(a.model, x, self.test_state)
        return distrib

    def observe_train(self, t, done, reset):
        ro = self.agent.rollout
        if self.prev_cols is None or self.prev_cols.t_cap != ro.cap:
            self.prev_cols, self.next_cols = StateColumns(ro.cap), StateColumns(ro.cap)
        self.prev_cols.write(t, self.train_prev_state)
        self.next_cols.write(t, self.train_state)
        self.train_prev_state = None
        ended = np.flatnonzero(done | reset)
        if len(ended):
            self.train_state = mask_recurrent_state_at(self.train_state, [int(i) for i in ended])

    def observe_eval(self, done, reset):
        ended = [i for i, (d, r) in enumerate(zip(done, reset)) if d or r]
        if ended:
            self.test_state = mask_recurrent_state_at(self.test_state, ended)

    # -- learning ----------------------------------------------------------------------------
    def _forward_packed(self, refs_col, seqs, cols, N):
        """The model on ``seqs`` packed longest-first, starting from the states stored in
        ``cols`` at each sequence's first position.  Returns ((distribs, values), positions in
        output order as a device tensor)."""
        a = self.agent
        order, flat, batch_sizes = packed_layout(seqs)
        first = np.array([seqs[i][0] for i in order], dtype=np.int64)
        (flat_dev, t_dev, e_dev) = [x.clone() for x in a._stage.upload([flat, first // N, first % N])]
        x = a._features(refs_col[flat_dev])
        rs = cols.pick(t_dev, e_dev)
        y, _ = a.model(PackedSequence(x, torch.from_numpy(batch_sizes)), rs)
        return unwrap_packed_sequences_recursive(y), flat_dev

    def update(self):
        a = self.agent
        ro = a.rollout
        T, N, k = ro.T, ro.N, ro.k
        dev = a.device
        n = T * N
        a._check_frames_alive(ro)
        on_dev = ro.d_action is not None
        up = a._stage.upload([
            ro.h_state[:T].reshape(n, k), ro.h_next[:T].reshape(n, k),
            (np.zeros(1, dtype=ro.h_action.dtype) if on_dev else
             ro.h_action[:T].reshape((n,) + ro.h_action.shape[2:])),
            ro.h_reward[:T].reshape(-1), ro.h_nonterm[:T].reshape(-1),
            a._cut_with_rollout_end(ro, T).reshape(-1)])
        s_refs, n_refs, actions, reward, nonterm, cut = [t.clone() for t in up]
        if on_dev:
            actions = ro.d_action[:T].reshape((n,) + tuple(ro.d_action.shape[2:])).clone()
        # the reference's ``memory``: finished fragments in completion order, then the open ones
        episodes = [np.arange(s, e + 1, dtype=np.int64) * N + env for env, s, e in ro.fragments()]
        assert sum(len(ep) for ep in episodes) == n
        # -- log pi(a | s), V(s), V(s') of every position (reference :56-107) ------------------
        log_probs = torch.empty(n, dtype=torch.float32, device=dev)
        v_pred = torch.empty(n, dtype=torch.float32, device=dev)
        next_v = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.no_grad(), evaluating(a.model):
            (distribs, vs), pos = self._forward_packed(s_refs, episodes, self.prev_cols, N)
            log_probs[pos] = distribs.log_prob(actions[pos]).float()
            v_pred[pos] = vs.reshape(-1).float()
            (_, nvs), pos = self._forward_packed(n_refs, episodes, self.next_cols, N)
            next_v[pos] = nvs.reshape(-1).float()
        # -- advantages: f64 throughout (mode 2), restarting at every fragment ------------------
        adv, v_teacher = ops.gae_scan(reward.view(T, N), v_pred.view(T, N), next_v.view(T, N),
                                      nonterm.view(T, N), cut.view(T, N), a.gamma, a.lambd, 2)
        adv, v_teacher = adv.view(-1), v_teacher.view(-1)
        dataset = (list(episodes) if a.max_recurrent_sequence_len is None else
                   _limit_sequence_length(episodes, a.max_recurrent_sequence_len))
        if a.obs_normalizer is not None:
            with torch.no_grad():
                a.obs_normalizer.experience(a._gather(s_refs))
        if a.standardize_advantages:
            from pfrl_amd.distributed import global_mean_std

            mean_std = global_mean_std(ops.adv_stats(adv), n)
        else:
            mean_std = None
        a._last_dataset = dict(adv=adv, v_teacher=v_teacher, v_pred=v_pred, log_prob=log_probs,
                               next_v_pred=next_v, mean_std=mean_std)
        cols = dict(s_refs=s_refs, actions=actions, adv=adv, v_teacher=v_teacher, v_pred=v_pred,
                    log_prob=log_probs, mean_std=mean_std)
        for _ in range(a.epochs):
            random.shuffle(dataset)
            for seqs in _yield_subset_of_sequences_with_fixed_number_of_items(dataset,
                                                                              a.minibatch_size):
                self._update_once(seqs, cols, N)
        with torch.no_grad():
            vart = torch.var(v_teacher, unbiased=False)
            ev = 1 - torch.var(v_teacher - v_pred, unbiased=False) / vart
            a.explained_variance = float("nan") if float(vart) == 0 else float(ev)

    def _update_once(self, seqs, c, N):
        """reference :534-606 for one group of sequences."""
        a = self.agent
        (distribs, vs_pred), pos = self._forward_packed(c["s_refs"], seqs, self.prev_cols, N)
        advs = c["adv"][pos]
        if a.standardize_advantages:
            advs = (advs - c["mean_std"][0]) / (c["mean_std"][1] + 1e-8)
        a.model.zero_grad()
        loss = a._lossfun(distribs.entropy(), vs_pred, distribs.log_prob(c["actions"][pos]),
                          vs_pred_old=c["v_pred"][pos][..., None], log_probs_old=c["log_prob"][pos],
                          advs=advs, vs_teacher=c["v_teacher"][pos][..., None])
        loss.backward()
        a.grad_reducer.all_reduce()
        if a.max_grad_norm is not None:
            clip_grad_norm_device_(a.model.parameters(), a.max_grad_norm)
        a.optimizer.step()
        a.n_updates += 1
