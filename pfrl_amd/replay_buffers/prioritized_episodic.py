"""Prioritized replay of whole episodes.

Mirrors ``pfrl.replay_buffers.PrioritizedEpisodicReplayBuffer``
(/root/reference/pfrl/replay_buffers/prioritized_episodic.py:9-77): the unit of prioritisation is
the EPISODE -- ``episodic_memory`` is a ``PrioritizedBuffer`` whose payloads are episodes -- while
``memory`` is a plain FIFO of single transitions bounded by ``capacity``; ``capacity_left`` counts
transitions and evicts whole episodes, oldest first, once it goes negative (:60-76).

The sum / min trees over the episodes are host trees by default (``collections.host_prioritized``:
one leaf per EPISODE, a few thousand at most) and the HBM-resident
``pfrl_amd.collections.PrioritizedBuffer`` when a ``device`` is given explicitly.  The episode
PAYLOADS follow ``EpisodicReplayBuffer``: on the host as lists of dicts by default, and in HBM
(transition table + frame ring, an episode = a run of consecutive entries, ragged gather by
pfrl_batch_episodes) once an agent with ``gpu >= 0`` has bound the buffer (``bind``).
"""
import collections

from pfrl_amd.collections.random_access_queue import RandomAccessQueue
from pfrl_amd.replay_buffer import random_subseq
from pfrl_amd.replay_buffers.episodic import EpisodicReplayBuffer
from pfrl_amd.replay_buffers.prioritized import PriorityWeightError


def _make_tree(wait_priority_after_sampling, device, max_episodes):
    if device is None:
        from pfrl_amd.collections.host_prioritized import HostPrioritizedBuffer

        return HostPrioritizedBuffer(capacity=None,
                                     wait_priority_after_sampling=wait_priority_after_sampling)
    from pfrl_amd.collections.prioritized import PrioritizedBuffer

    return PrioritizedBuffer(capacity=None, device=device, max_size=max_episodes,
                             wait_priority_after_sampling=wait_priority_after_sampling)


class PrioritizedEpisodicReplayBuffer(EpisodicReplayBuffer, PriorityWeightError):
    def __init__(self, capacity=None, alpha=0.6, beta0=0.4, betasteps=2e5, eps=1e-8,
                 normalize_by_max=True, default_priority_func=None, uniform_ratio=0,
                 wait_priority_after_sampling=True, return_sample_weights=True,
                 error_min=None, error_max=None, device=None, max_episodes=1 << 20,
                 _tree_factory=_make_tree):
        self.current_episode = collections.defaultdict(list)
        self.episodic_memory = _tree_factory(wait_priority_after_sampling, device, max_episodes)
        self.memory = RandomAccessQueue(maxlen=capacity)
        self.capacity = capacity
        self._device_opts = dict(max_size=None, slack=None, frame_slots=None)
        self.capacity_left = capacity
        self.default_priority_func = default_priority_func
        self.uniform_ratio = uniform_ratio
        self.return_sample_weights = return_sample_weights
        PriorityWeightError.__init__(self, alpha, beta0, betasteps, eps, normalize_by_max,
                                     error_min=error_min, error_max=error_max)

    def _commit(self, episode):
        if self.store is not None:
            # device payloads: the episode's transitions become consecutive entries
            from pfrl_amd.replay_buffer import DeviceEpisode
            from pfrl_amd.replay_buffers.episodic import _EpisodeRef

            st = self.store
            first = st.n_entries
            for tid in episode:
                st.add_entry([tid], span_check=False)
            ref = _EpisodeRef(first, len(episode))
            self._note_commit(episode)
            priority = None
            if self.default_priority_func is not None:
                priority = self.default_priority_func(DeviceEpisode(st, first, len(episode)))
            self.episodic_memory.append(ref, priority=priority)
            if self.capacity is not None:
                # ``memory`` is a FIFO of single transitions with maxlen = capacity (reference
                # :47): the oldest TRANSITIONS leave one by one, whatever the episodes do
                self.memory.head = max(self.memory.head, st.n_entries - self.capacity)
            if self.capacity_left is None:
                # unbounded: nothing is ever evicted, the entry ring is the limit
                if st.n_entries > st.bound:
                    raise RuntimeError("unbounded PrioritizedEpisodicReplayBuffer exceeded its device "
                                       "allocation (max_size=%d)" % st.bound)
                return
            self.capacity_left -= len(ref)
            while self.capacity_left < 0:
                self.capacity_left += len(self.episodic_memory.popleft())
                self._note_evict()
            return
        priority = None
        if self.default_priority_func is not None:
            priority = self.default_priority_func(episode)
        self.memory.extend(episode)
        self.episodic_memory.append(episode, priority=priority)
        if self.capacity_left is None:
            return
        self.capacity_left -= len(episode)
        while self.capacity_left < 0:
            self.capacity_left += len(self.episodic_memory.popleft())

    def sample_episodes(self, n_episodes, max_len=None):
        assert len(self.episodic_memory) >= n_episodes
        episodes, probabilities, min_prob = self.episodic_memory.sample(
            n_episodes, uniform_ratio=self.uniform_ratio)
        if self.store is not None:
            from pfrl_amd.replay_buffer import DeviceEpisode

            episodes = [DeviceEpisode(self.store, ep.first, ep.length) for ep in episodes]
        if max_len is not None:
            episodes = [random_subseq(ep, max_len) for ep in episodes]
        if not self.return_sample_weights:
            return episodes
        return episodes, self.weights_from_probabilities(probabilities, min_prob)

    def update_errors(self, errors):
        self.episodic_memory.set_last_priority(self.priority_from_errors(errors))
