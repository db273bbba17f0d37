"""Replay-buffer interface, minibatch assembly and update schedule.

Mirrors ``pfrl.replay_buffer`` (/root/reference/pfrl/replay_buffer.py):
``AbstractReplayBuffer`` (:15-114), ``batch_experiences`` (:157-212) and
``ReplayUpdater`` (:290-356) with identical signatures and error behaviour.
``batch_experiences`` has two entry forms:

* a :class:`DeviceExperienceBatch` (what the device buffers' ``sample`` returns)
  -> ONE fused HIP launch (pfrl_batch_experiences) does the n-step collapse and
  both observation gathers in HBM;
* a plain list of lists of transition dicts (host data, e.g. the CPU ``gpu=-1``
  path) -> vectorised with torch exactly as the reference does.
"""
from abc import ABCMeta, abstractmethod

import numpy as np
import torch

from pfrl_amd.utils.batch_states import batch_states


class AbstractReplayBuffer(object, metaclass=ABCMeta):
    """Common interface of replay buffers (reference :15-114)."""

    @abstractmethod
    def append(self, state, action, reward, next_state=None, next_action=None,
               is_state_terminal=False, env_id=0, **kwargs):
        raise NotImplementedError

    @abstractmethod
    def sample(self, n):
        raise NotImplementedError

    @abstractmethod
    def __len__(self):
        raise NotImplementedError

    @abstractmethod
    def save(self, filename):
        raise NotImplementedError

    @abstractmethod
    def load(self, filename):
        raise NotImplementedError

    @property
    @abstractmethod
    def capacity(self):
        raise NotImplementedError

    @abstractmethod
    def stop_current_episode(self, env_id=0):
        raise NotImplementedError


class AbstractEpisodicReplayBuffer(AbstractReplayBuffer):
    """Interface marker kept for isinstance checks; episodic (recurrent)
    replay is outside the hot path."""

    @abstractmethod
    def sample_episodes(self, n_episodes, max_len=None):
        raise NotImplementedError

    @property
    @abstractmethod
    def n_episodes(self):
        raise NotImplementedError


class DeviceExperienceBatch:
    """What device-resident buffers return from ``sample(n)``.

    Holds the sampled entry-ring slots on the device.  For API compatibility
    it is also a lazy ``Sequence[Sequence[Mapping]]`` (reference: list of
    n-step lists of transition dicts); indexing it materialises host views."""

    def __init__(self, store, slots_dev, seqs, weights_dev=None):
        self.store = store
        self.slots_dev = slots_dev
        self.seqs = seqs
        self.weights_dev = weights_dev
        self._weights_host = None

    def __len__(self):
        return len(self.seqs)

    def __getitem__(self, b):
        if isinstance(b, slice):
            raise TypeError("slice a DeviceExperienceBatch on the device instead")
        w = None
        if self.weights_dev is not None:
            if self._weights_host is None:
                self._weights_host = self.weights_dev.cpu().numpy()
            w = self._weights_host[b]
        return self.store.entry_view(int(self.seqs[b]), w)

    def __iter__(self):
        for b in range(len(self)):
            yield self[b]

    @property
    def has_weight(self):
        return self.weights_dev is not None


def batch_experiences(experiences, device, phi, gamma, batch_states=batch_states):
    """Vectorise k sampled n-step experiences (reference :157-212).

    Returns a dict with state, action, reward, next_state, is_state_terminal,
    discount (and next_action when every last transition has one)."""
    if isinstance(experiences, DeviceExperienceBatch):
        return experiences.store.fetch(experiences, phi, gamma)

    first = [e[0] for e in experiences]
    last = [e[-1] for e in experiences]
    rewards = []
    terminals = []
    for e in experiences:
        acc = 0
        flag = False
        for i, tr in enumerate(e):
            acc = acc + (gamma ** i) * tr["reward"]
            flag = flag or bool(tr["is_state_terminal"])
        rewards.append(acc)
        terminals.append(flag)
    def as_tensor(values):
        # array-valued entries (continuous actions): stack on the host first -- one
        # conversion instead of torch's element-wise walk over a list of ndarrays
        if len(values) and isinstance(values[0], np.ndarray):
            values = np.stack(values)
        return torch.as_tensor(values, device=device)

    out = {
        "state": batch_states([t["state"] for t in first], device, phi),
        "action": as_tensor([t["action"] for t in first]),
        "reward": torch.as_tensor(rewards, dtype=torch.float32, device=device),
        "next_state": batch_states([t["next_state"] for t in last], device, phi),
        "is_state_terminal": torch.as_tensor(terminals, dtype=torch.float32, device=device),
        "discount": torch.as_tensor([gamma ** len(e) for e in experiences], dtype=torch.float32,
                                    device=device),
    }
    if all(t["next_action"] is not None for t in last):
        out["next_action"] = as_tensor([t["next_action"] for t in last])
    return out


def random_subseq(seq, subseq_len):
    """A uniformly placed window of ``subseq_len`` items, or all of ``seq`` when it is not longer
    (reference :149-154; one ``np.random.randint`` draw only when a window is cut)."""
    excess = len(seq) - subseq_len
    if excess <= 0:
        return seq
    start = np.random.randint(0, excess + 1)
    return seq[start:start + subseq_len]


class DeviceEpisode:
    """A sampled episode (or a ``random_subseq`` window of one) of a device-resident
    EpisodicReplayBuffer: ``length`` consecutive one-transition entries of the entry ring from
    sequence number ``first``.  Behaves as the list of transition dicts the reference returns
    (``len``, indexing, slicing -- views built from the host mirrors, observations as
    ``DeviceObs``); ``batch_recurrent_experiences`` never looks at them: it hands the (first,
    length) pairs to ONE ragged gather launch (pfrl_batch_episodes)."""

    __slots__ = ("store", "first", "length")

    def __init__(self, store, first, length):
        self.store, self.first, self.length = store, int(first), int(length)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        if isinstance(i, slice):
            start, stop, step = i.indices(self.length)
            assert step == 1
            return DeviceEpisode(self.store, self.first + start, max(0, stop - start))
        if i < 0:
            i += self.length
        if not 0 <= i < self.length:
            raise IndexError("episode index out of range")
        return self.store.entry_view(self.first + i)[0]

    def __iter__(self):
        return (self[i] for i in range(self.length))


def _batch_device_episodes(experiences, device, phi, gamma):
    from pfrl_amd.utils.recurrent import concatenate_recurrent_states, recurrent_state_from_numpy

    store = experiences[0].store
    out = store.fetch_episodes([(ep.first, ep.length) for ep in experiences], phi, gamma)

    def initial_state(key):
        return recurrent_state_from_numpy(
            concatenate_recurrent_states([ep[0][key] for ep in experiences]), device)

    out["recurrent_state"] = initial_state("recurrent_state")
    out["next_recurrent_state"] = initial_state("next_recurrent_state")
    # `next_action` (SARSA-style agents) lives in the store's per-transition extras; emitted, as
    # the reference does (:283-285), only when every transition has one -- in packed time-major order
    table = (store.h_extra or {}).get("next_action")
    if table:
        from pfrl_amd.utils.recurrent import flatten_sequences_time_first

        tids = flatten_sequences_time_first(
            [[int(store.h_e_tids[(ep.first + i) % store.E][0]) for i in range(ep.length)]
             for ep in experiences])
        if all(t in table for t in tids):
            out["next_action"] = torch.as_tensor([table[t] for t in tids], device=device)
    return out


_DEFAULT_BATCH_STATES = batch_states


def batch_recurrent_experiences(experiences, device, phi, gamma, batch_states=batch_states):
    """Vectorise sampled episodes for a recurrent update (reference :219-287).

    ``experiences`` is a list of episodes (lists of transition dicts) sorted by descending length,
    as ``pack_sequence`` needs.  ``state`` / ``next_state`` stay per-episode lists of
    ``(len, ...)`` batches; the per-transition columns are flat in packed (time-major) order; the
    recurrent states are those stored with each episode's FIRST transition, stacked on axis 1."""
    from pfrl_amd.utils.recurrent import (concatenate_recurrent_states,
                                          flatten_sequences_time_first,
                                          recurrent_state_from_numpy)

    lengths = [len(ep) for ep in experiences]
    assert all(a >= b for a, b in zip(lengths, lengths[1:])), "episodes must be sorted by length"
    if (experiences and all(isinstance(ep, DeviceEpisode) for ep in experiences)
            and batch_states is _DEFAULT_BATCH_STATES):
        # episode payloads live in HBM: one ragged gather launch, nothing crosses PCIe but
        # three small index arrays
        return _batch_device_episodes(experiences, device, phi, gamma)
    flat = flatten_sequences_time_first(experiences)

    def column(key, **kw):
        return torch.as_tensor([tr[key] for tr in flat], device=device, **kw)

    def initial_state(key):
        return recurrent_state_from_numpy(
            concatenate_recurrent_states([ep[0][key] for ep in experiences]), device)

    out = {
        "state": [batch_states([tr["state"] for tr in ep], device, phi) for ep in experiences],
        "action": column("action"),
        "reward": column("reward", dtype=torch.float),
        "next_state": [batch_states([tr["next_state"] for tr in ep], device, phi)
                       for ep in experiences],
        "is_state_terminal": column("is_state_terminal", dtype=torch.float),
        "discount": torch.full((len(flat),), gamma, dtype=torch.float, device=device),
        "recurrent_state": initial_state("recurrent_state"),
        "next_recurrent_state": initial_state("next_recurrent_state"),
    }
    if all(tr["next_action"] is not None for tr in flat):
        out["next_action"] = column("next_action")
    return out


class ReplayUpdater(object):
    """Update schedule (reference :290-356): skip until ``replay_start_size``
    transitions are stored, then every ``update_interval`` steps draw
    ``n_times_update`` minibatches and hand each to ``update_func``."""

    def __init__(self, replay_buffer, update_func, batchsize, episodic_update, n_times_update,
                 replay_start_size, update_interval, episodic_update_len=None):
        assert batchsize <= replay_start_size
        self.replay_buffer = replay_buffer
        self.update_func = update_func
        self.batchsize = batchsize
        self.episodic_update = episodic_update
        self.episodic_update_len = episodic_update_len
        self.n_times_update = n_times_update
        self.replay_start_size = replay_start_size
        self.update_interval = update_interval

    def update_if_necessary(self, iteration):
        if len(self.replay_buffer) < self.replay_start_size:
            return False
        if self.episodic_update and self.replay_buffer.n_episodes < self.batchsize:
            return False
        if iteration % self.update_interval != 0:
            return False
        for _ in range(self.n_times_update):
            if self.episodic_update:
                episodes = self.replay_buffer.sample_episodes(self.batchsize,
                                                              self.episodic_update_len)
                self.update_func(episodes)
            else:
                self.update_func(self.replay_buffer.sample(self.batchsize))
        return True
