"""PPO on the host: the ``gpu=None / -1`` plumbing path, including recurrent models.

``pfrl_amd.agents.PPO`` keeps its rollout in HBM and drives the HIP kernels when it is given a
GPU.  Without one it hands acting, observing and updating to :class:`HostRollouts` below, which
restates the reference's list-of-dicts algorithm (/root/reference/pfrl/agents/ppo.py: dataset
helpers :36-257, acting / observing :684-807, updates :465-632) with stock torch ops.  The
module-level helpers keep the reference's names because its tests and users call them directly.

Episodes are lists of transition dicts ``{state, action, reward, next_state, nonterminal}``
(+ ``recurrent_state`` / ``next_recurrent_state`` for recurrent models) and gain ``log_prob``,
``v_pred``, ``next_v_pred``, ``adv`` and ``v_teacher`` when a dataset is made.  An "episode" here
is a fragment: it also ends at a reset and at the rollout boundary, and the advantage scan
restarts at every fragment.
"""
import itertools
import random

import numpy as np
import torch

from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.mode_of_distribution import mode_of_distribution
from pfrl_amd.utils.recurrent import (concatenate_recurrent_states, flatten_sequences_time_first,
                                      get_recurrent_state_at, mask_recurrent_state_at,
                                      one_step_forward, pack_and_forward)


# ---- dataset helpers ---------------------------------------------------------------------------
def _add_advantage_and_value_target_to_episode(episode, gamma, lambd):
    """Generalised advantage estimation, scanned backwards over one fragment (reference :36-47).
    The arithmetic type follows the stored scalars (Python / NumPy promotion)."""
    adv = 0.0
    for tr in reversed(episode):
        delta = tr["reward"] + gamma * tr["nonterminal"] * tr["next_v_pred"] - tr["v_pred"]
        adv = delta + gamma * lambd * adv
        tr["adv"] = adv
        tr["v_teacher"] = adv + tr["v_pred"]


def _add_advantage_and_value_target_to_episodes(episodes, gamma, lambd):
    for episode in episodes:
        _add_advantage_and_value_target_to_episode(episode, gamma=gamma, lambd=lambd)


def _actions(transitions, device):
    """Stored actions as one tensor (stacked on the host first: array-valued actions would
    otherwise be converted element by element)."""
    return torch.as_tensor(np.asarray([tr["action"] for tr in transitions]), device=device)


def _states(transitions, key, batch_states, device, phi, obs_normalizer):
    x = batch_states([tr[key] for tr in transitions], device, phi)
    return obs_normalizer(x, update=False) if obs_normalizer else x


def _add_log_prob_and_value_to_episodes(episodes, model, phi, batch_states, obs_normalizer,
                                        device):
    """One pass over all states and one over all next-states of the rollout (reference :110-142);
    results are scattered back as NumPy scalars."""
    dataset = list(itertools.chain.from_iterable(episodes))
    with torch.no_grad(), evaluating(model):
        distribs, vs = model(_states(dataset, "state", batch_states, device, phi, obs_normalizer))
        _, next_vs = model(_states(dataset, "next_state", batch_states, device, phi,
                                   obs_normalizer))
        actions = _actions(dataset, device)
        columns = (distribs.log_prob(actions).cpu().numpy(), vs.cpu().numpy().ravel(),
                   next_vs.cpu().numpy().ravel())
    for tr, log_prob, v, next_v in zip(dataset, *columns):
        tr["log_prob"], tr["v_pred"], tr["next_v_pred"] = log_prob, v, next_v


def _add_log_prob_and_value_to_episodes_recurrent(episodes, model, phi, batch_states,
                                                  obs_normalizer, device):
    """The same for a recurrent model: fragments are packed longest-first and each starts from
    the recurrent state stored with its first transition (reference :56-107)."""
    episodes = sorted(episodes, key=len, reverse=True)
    flat = flatten_sequences_time_first(episodes)
    with torch.no_grad(), evaluating(model):
        def run(state_key, rs_key):
            seqs = [_states(ep, state_key, batch_states, device, phi, obs_normalizer)
                    for ep in episodes]
            rs = concatenate_recurrent_states([ep[0][rs_key] for ep in episodes])
            return pack_and_forward(model, seqs, rs)[0]

        distribs, vs = run("state", "recurrent_state")
        _, next_vs = run("next_state", "next_recurrent_state")
        actions = _actions(flat, device)
        columns = (distribs.log_prob(actions).cpu().numpy(), vs.cpu().numpy(),
                   next_vs.cpu().numpy())
    for tr, log_prob, v, next_v in zip(flat, *columns):
        tr["log_prob"], tr["v_pred"], tr["next_v_pred"] = float(log_prob), float(v), float(next_v)


def _limit_sequence_length(sequences, max_len):
    """Cut every sequence into consecutive chunks of at most ``max_len`` items (reference :145-155)."""
    assert max_len > 0
    return [seq[i:i + max_len] for seq in sequences for i in range(0, len(seq), max_len)]


def _yield_subset_of_sequences_with_fixed_number_of_items(sequences, n_items):
    """Walk the sequences in order and yield groups holding exactly ``n_items`` items, splitting a
    sequence where a group fills up; a final group that cannot be filled is dropped
    (reference :158-181)."""
    assert n_items > 0
    group, room = [], n_items
    pending = list(reversed(sequences))
    while pending:
        seq = pending.pop()
        if len(seq) > room:
            seq, rest = seq[:room], seq[room:]
            pending.append(rest)
        group.append(seq)
        room -= len(seq)
        if room == 0:
            yield group
            group, room = [], n_items


def _compute_explained_variance(transitions):
    """1 - Var[return - v] / Var[return] (reference :184-197)."""
    returns = np.array([tr["v_teacher"] for tr in transitions])
    values = np.array([tr["v_pred"] for tr in transitions])
    var = np.var(returns)
    return np.nan if var == 0 else float(1 - np.var(returns - values) / var)


def _make_dataset_recurrent(episodes, model, phi, batch_states, obs_normalizer, gamma, lambd,
                            max_recurrent_sequence_len, device):
    """A list of sequences ready for recurrent updates (reference :200-231)."""
    _add_log_prob_and_value_to_episodes_recurrent(
        episodes=episodes, model=model, phi=phi, batch_states=batch_states,
        obs_normalizer=obs_normalizer, device=device)
    _add_advantage_and_value_target_to_episodes(episodes, gamma=gamma, lambd=lambd)
    if max_recurrent_sequence_len is None:
        return list(episodes)
    return _limit_sequence_length(episodes, max_recurrent_sequence_len)


def _make_dataset(episodes, model, phi, batch_states, obs_normalizer, gamma, lambd, device):
    """A flat list of transitions ready for updates (reference :234-248)."""
    _add_log_prob_and_value_to_episodes(
        episodes=episodes, model=model, phi=phi, batch_states=batch_states,
        obs_normalizer=obs_normalizer, device=device)
    _add_advantage_and_value_target_to_episodes(episodes, gamma=gamma, lambd=lambd)
    return list(itertools.chain.from_iterable(episodes))


def _yield_minibatches(dataset, minibatch_size, num_epochs):
    """``num_epochs`` passes in minibatches taken from the tail of a buffer that is refilled, at
    the front, with fresh ``random.sample`` permutations of the dataset (reference :251-262)."""
    assert dataset
    from pfrl_amd.agents.ppo import _yield_minibatch_positions

    for positions in _yield_minibatch_positions(len(dataset), minibatch_size, num_epochs):
        yield [dataset[i] for i in positions]


# ---- the host side of the agent ----------------------------------------------------------------
def _ended(batch_done, batch_reset):
    return [i for i, (d, r) in enumerate(zip(batch_done, batch_reset)) if d or r]


class HostRollouts:
    """Acting, observing and updating of a :class:`~pfrl_amd.agents.PPO` created without a GPU.
    Hyper-parameters, model, optimizer and statistics windows are the agent's."""

    def __init__(self, agent):
        self.agent = agent
        self.memory = []                    # finished fragments of this rollout
        self.batch_last_episode = None      # per env: the fragment in progress
        self.batch_last_state = None
        self.batch_last_action = None
        self.train_recurrent_states = None
        self.train_prev_recurrent_states = None
        self.test_recurrent_states = None

    # -- acting ----------------------------------------------------------------------------
    def _input(self, batch_obs):
        a = self.agent
        x = a.batch_states(batch_obs, a.device, a.phi)
        return a.obs_normalizer(x, update=False) if a.obs_normalizer else x

    def batch_act_train(self, batch_obs):
        a = self.agent
        n = len(batch_obs)
        if self.batch_last_episode is None:
            self.batch_last_episode = [[] for _ in range(n)]
            self.batch_last_state = [None] * n
            self.batch_last_action = [None] * n
        assert len(self.batch_last_episode) == n
        with torch.no_grad(), evaluating(a.model):
            if a.recurrent:
                assert self.train_prev_recurrent_states is None
                self.train_prev_recurrent_states = self.train_recurrent_states
                (distrib, value), self.train_recurrent_states = one_step_forward(
                    a.model, self._input(batch_obs), self.train_prev_recurrent_states)
            else:
                distrib, value = a.model(self._input(batch_obs))
            batch_action = a._sample_action(distrib).cpu().numpy()
            a.entropy_record.extend(distrib.entropy())
            a.value_record.extend(value)
        self.batch_last_state = list(batch_obs)
        self.batch_last_action = list(batch_action)
        return batch_action

    def batch_act_eval(self, batch_obs):
        a = self.agent
        with torch.no_grad(), evaluating(a.model):
            if a.recurrent:
                (distrib, _), self.test_recurrent_states = one_step_forward(
                    a.model, self._input(batch_obs), self.test_recurrent_states)
            else:
                distrib, _ = a.model(self._input(batch_obs))
            action = mode_of_distribution(distrib) if a.act_deterministically else distrib.sample()
        return action.cpu().numpy()

    # -- observing ---------------------------------------------------------------------------
    def batch_observe_train(self, batch_obs, batch_reward, batch_done, batch_reset):
        a = self.agent
        for i in range(len(batch_obs)):
            state = self.batch_last_state[i]
            if state is not None:
                assert self.batch_last_action[i] is not None
                tr = dict(state=state, action=self.batch_last_action[i], reward=batch_reward[i],
                          next_state=batch_obs[i], nonterminal=0.0 if batch_done[i] else 1.0)
                if a.recurrent:
                    tr["recurrent_state"] = get_recurrent_state_at(
                        self.train_prev_recurrent_states, i, detach=True)
                    tr["next_recurrent_state"] = get_recurrent_state_at(
                        self.train_recurrent_states, i, detach=True)
                self.batch_last_episode[i].append(tr)
            if batch_done[i] or batch_reset[i]:
                assert self.batch_last_episode[i]
                self.memory.append(self.batch_last_episode[i])
                self.batch_last_episode[i] = []
            self.batch_last_state[i] = None
            self.batch_last_action[i] = None
        self.train_prev_recurrent_states = None
        if a.recurrent:
            ended = _ended(batch_done, batch_reset)
            if ended:
                self.train_recurrent_states = mask_recurrent_state_at(
                    self.train_recurrent_states, ended)
        self.update_if_dataset_is_ready()

    def batch_observe_eval(self, batch_obs, batch_reward, batch_done, batch_reset):
        if self.agent.recurrent:
            ended = _ended(batch_done, batch_reset)
            if ended:
                self.test_recurrent_states = mask_recurrent_state_at(
                    self.test_recurrent_states, ended)

    # -- learning ----------------------------------------------------------------------------
    def _dataset_size(self):
        open_fragments = self.batch_last_episode or ()
        return sum(len(ep) for ep in itertools.chain(self.memory, open_fragments))

    def update_if_dataset_is_ready(self):
        a = self.agent
        size = self._dataset_size()
        if size < a.update_interval:
            return
        # the fragments in progress are cut at the rollout boundary (reference :448-456)
        for i, fragment in enumerate(self.batch_last_episode or ()):
            if fragment:
                self.memory.append(fragment)
                self.batch_last_episode[i] = []
        common = dict(episodes=self.memory, model=a.model, phi=a.phi, batch_states=a.batch_states,
                      obs_normalizer=a.obs_normalizer, gamma=a.gamma, lambd=a.lambd,
                      device=a.device)
        if a.recurrent:
            self._update_recurrent(_make_dataset_recurrent(
                max_recurrent_sequence_len=a.max_recurrent_sequence_len, **common))
        else:
            dataset = _make_dataset(**common)
            assert len(dataset) == size
            self._update(dataset)
        a.explained_variance = _compute_explained_variance(
            list(itertools.chain.from_iterable(self.memory)))
        self.memory = []

    def _advantage_statistics(self, transitions):
        a = self.agent
        if a.obs_normalizer:
            a.obs_normalizer.experience(a.batch_states([tr["state"] for tr in transitions],
                                                       a.device, a.phi))
        if not a.standardize_advantages:
            return None, None
        advs = torch.tensor([tr["adv"] for tr in transitions], device=a.device)
        std, mean = torch.std_mean(advs, unbiased=False)
        # env-sharded data parallelism: statistics of the union of the ranks' rollouts
        # (no-op for a single process)
        from pfrl_amd.distributed import global_mean_std

        mean, std = global_mean_std(torch.stack([mean, std]), advs.numel())
        return mean, std

    def _column(self, transitions, key, column=False):
        values = [[tr[key]] for tr in transitions] if column else [tr[key] for tr in transitions]
        return torch.tensor(values, dtype=torch.float, device=self.agent.device)

    def _step(self, transitions, distribs, vs_pred, mean_advs, std_advs):
        """Loss, backward, clip, optimizer step for one minibatch given the fresh model outputs
        for ``transitions`` (flat, in the order the outputs are in)."""
        a = self.agent
        actions = _actions(transitions, a.device)
        advs = self._column(transitions, "adv")
        if a.standardize_advantages:
            advs = (advs - mean_advs) / (std_advs + 1e-8)
        a.model.zero_grad()
        loss = a._lossfun(distribs.entropy(), vs_pred, distribs.log_prob(actions),
                          vs_pred_old=self._column(transitions, "v_pred", column=True),
                          log_probs_old=self._column(transitions, "log_prob"), advs=advs,
                          vs_teacher=self._column(transitions, "v_teacher", column=True))
        loss.backward()
        a.grad_reducer.all_reduce()
        if a.max_grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(a.model.parameters(), a.max_grad_norm)
        a.optimizer.step()
        a.n_updates += 1

    def _update(self, dataset):
        a = self.agent
        mean_advs, std_advs = self._advantage_statistics(dataset)
        for batch in _yield_minibatches(dataset, minibatch_size=a.minibatch_size,
                                        num_epochs=a.epochs):
            distribs, vs_pred = a.model(_states(batch, "state", a.batch_states, a.device, a.phi,
                                                a.obs_normalizer))
            self._step(batch, distribs, vs_pred, mean_advs, std_advs)

    def _update_recurrent(self, dataset):
        a = self.agent
        mean_advs, std_advs = self._advantage_statistics(
            list(itertools.chain.from_iterable(dataset)))
        for _ in range(a.epochs):
            random.shuffle(dataset)
            for sequences in _yield_subset_of_sequences_with_fixed_number_of_items(
                    dataset, a.minibatch_size):
                self._update_once_recurrent(sequences, mean_advs, std_advs)

    def _update_once_recurrent(self, episodes, mean_advs, std_advs):
        a = self.agent
        assert std_advs is None or std_advs > 0
        episodes = sorted(episodes, key=len, reverse=True)
        seqs = [_states(ep, "state", a.batch_states, a.device, a.phi, a.obs_normalizer)
                for ep in episodes]
        rs = concatenate_recurrent_states([ep[0]["recurrent_state"] for ep in episodes])
        (distribs, vs_pred), _ = pack_and_forward(a.model, seqs, rs)
        self._step(flatten_sequences_time_first(episodes), distribs, vs_pred, mean_advs, std_advs)
