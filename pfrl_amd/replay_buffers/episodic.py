"""Replay of whole episodes, for recurrent models.

Mirrors ``pfrl.replay_buffers.EpisodicReplayBuffer``
(/root/reference/pfrl/replay_buffers/episodic.py:9-98): transitions of the episode in progress
are held per ``env_id`` and become visible -- as one item of ``episodic_memory`` and as
``len(episode)`` single-transition items of ``memory`` -- only when the episode ends (terminal
transition or ``stop_current_episode``).  ``capacity`` bounds the number of TRANSITIONS; whole
episodes are evicted oldest-first until it is respected (:91-97), so ``len(buffer)`` never exceeds
it after a commit.

Storage back-ends, as for the flat buffers:

* device (``device='cuda:N'`` or bound by an agent created with ``gpu>=0``): SURVEY.md 8(f) row 4.
  Transitions go into the HBM transition table / frame ring of a :class:`DeviceReplayStore` as
  they arrive; when an episode ends its transitions become CONSECUTIVE one-transition entries of
  the entry ring, so an episode is the pair (first entry, length) -- host integers, like every
  other piece of bookkeeping.  ``sample_episodes`` returns :class:`DeviceEpisode` windows and
  ``batch_recurrent_experiences`` turns them into packed minibatches with ONE ragged gather launch
  (pfrl_batch_episodes): episode payloads never leave HBM.  Sampling consumes the global NumPy
  stream draw-for-draw (``sample_n_k`` for the episode choice, then one ``randint`` per episode
  that is cut to ``max_len``).
* host (no device, or ``gpu=None/-1``): episodes are lists of the caller's transition dicts, held
  by reference exactly as the reference does.
"""
import collections
import pickle

from pfrl_amd.collections.random_access_queue import RandomAccessQueue
from pfrl_amd.replay_buffer import AbstractEpisodicReplayBuffer, random_subseq


class EpisodicReplayBuffer(AbstractEpisodicReplayBuffer):
    capacity = None
    # (class-level defaults: subclasses with their own __init__ start on the host back-end)
    store = None
    device = None
    _bound = False
    _device_opts = dict(max_size=None, slack=None, frame_slots=None)

    def __init__(self, capacity=None, device=None, max_size=None, slack=None, frame_slots=None):
        self.capacity = capacity
        self.current_episode = collections.defaultdict(list)
        self.episodic_memory = RandomAccessQueue()
        self.memory = RandomAccessQueue()
        self._device_opts = dict(max_size=max_size, slack=slack, frame_slots=frame_slots)
        self.device = None
        self.store = None
        self._bound = False
        if device is not None:
            self.bind(device)

    # -- back-end selection (same protocol as ReplayBuffer.bind) -------------------------------
    def bind(self, device, phi=None):
        import torch

        device = torch.device(device)
        if self._bound:
            if self.device != device:
                raise RuntimeError("replay buffer already bound to %s" % self.device)
            if self.store is not None and phi is not None:
                self.store.set_phi(phi)
            return self
        assert len(self.memory) == 0 and not any(self.current_episode.values()), \
            "bind() before the first append"
        self.device, self._bound = device, True
        if device.type == "cuda":
            from pfrl_amd.replay_buffers.device_replay import DeviceReplayStore
            from pfrl_amd.replay_buffers.replay_buffer import _DeviceQueue

            self.store = DeviceReplayStore(device, self.capacity, 1, **self._device_opts)
            if phi is not None:
                self.store.set_phi(phi)
            # FIFO of one-transition entries whose head moves by whole episodes (below)
            self.memory = _DeviceQueue(self.store, None)
        return self

    @property
    def is_device(self):
        return self.store is not None

    # -- ingest ------------------------------------------------------------------------------
    def append(self, state, action, reward, next_state=None, next_action=None,
               is_state_terminal=False, env_id=0, **kwargs):
        self._bound = True
        if self.store is not None:
            if next_action is not None:
                kwargs = dict(kwargs, next_action=next_action)
            self._guard_ring()
            item = self.store.add_transition(state, action, reward, next_state, is_state_terminal,
                                             kwargs or None)
        else:
            item = dict(state=state, action=action, reward=reward, next_state=next_state,
                        next_action=next_action, is_state_terminal=is_state_terminal, **kwargs)
        self.current_episode[env_id].append(item)
        if is_state_terminal:
            self.stop_current_episode(env_id=env_id)

    # -- device back-end: the transition ring vs the episodes that are still alive ---------------
    # An episode's rows sit at tid % R, interleaved with the rows of every other env, from its first
    # step until it is evicted -- a span that is NOT bounded by the n-step slack of the flat buffers
    # (n_envs x episode length + capacity).  `_live_first` holds the first tid of every committed
    # episode in commit (= eviction) order; the oldest live tid only ever grows, so it is cached
    # and recomputed when the ring catches up with it.
    _live_first = None
    _oldest_live = 0

    def _guard_ring(self):
        """Called before transition ``n_trans`` is written: the row it overwrites
        (``n_trans - R``) must not belong to a live (committed or running) episode."""
        st = self.store
        dead = st.n_trans - st.R
        if dead < self._oldest_live:
            return
        cands = [ep[0] for ep in self.current_episode.values() if ep]
        if self._live_first:
            cands.append(min(self._live_first))
        self._oldest_live = min(cands) if cands else st.n_trans
        if dead >= self._oldest_live:
            raise RuntimeError(
                "episodic replay: the device transition ring (%d rows = capacity %s + slack %d) "
                "would overwrite a live episode (oldest live transition %d, next %d): episodes of "
                "all envs interleave in the ring, so it must hold capacity + n_envs x episode "
                "length rows -- pass a larger `slack` (or `max_size` for an unbounded buffer)"
                % (st.R, self.capacity, st.slack, self._oldest_live, st.n_trans))

    def _note_commit(self, tids):
        if self._live_first is None:
            self._live_first = collections.deque()
        self._live_first.append(tids[0])

    def _note_evict(self):
        if self._live_first:
            self._live_first.popleft()

    def stop_current_episode(self, env_id=0):
        episode = self.current_episode[env_id]
        if not episode:
            return
        self.current_episode[env_id] = []
        self._commit(episode)

    def _commit(self, episode):
        if self.store is not None:
            return self._commit_device(episode)
        self.episodic_memory.append(episode)
        for transition in episode:
            self.memory.append([transition])
        if self.capacity is None:
            return
        while len(self.memory) > self.capacity:
            for _ in self.episodic_memory.popleft():
                self.memory.popleft()

    def _commit_device(self, tids):
        """The episode's transitions become consecutive entries: (first entry seq, length)."""
        st = self.store
        first = st.n_entries
        for tid in tids:
            st.add_entry([tid], span_check=False)
        if st.n_entries - self.memory.head > st.bound and self.capacity is None:
            raise RuntimeError("unbounded EpisodicReplayBuffer exceeded its device allocation "
                               "(max_size=%d)" % st.bound)
        self.episodic_memory.append(_EpisodeRef(first, len(tids)))
        self._note_commit(tids)
        if self.capacity is None:
            return
        while len(self.memory) > self.capacity:
            self.memory.head += len(self.episodic_memory.popleft())    # whole episodes leave
            self._note_evict()

    # -- sampling ----------------------------------------------------------------------------
    def sample(self, n):
        assert len(self.memory) >= n
        return self.memory.sample(n)

    def sample_episodes(self, n_episodes, max_len=None):
        assert len(self.episodic_memory) >= n_episodes
        episodes = self.episodic_memory.sample(n_episodes)
        if self.store is not None:
            from pfrl_amd.replay_buffer import DeviceEpisode

            episodes = [DeviceEpisode(self.store, ep.first, ep.length) for ep in episodes]
        if max_len is None:
            return episodes
        return [random_subseq(ep, max_len) for ep in episodes]

    def __len__(self):
        return len(self.memory)

    @property
    def n_episodes(self):
        return len(self.episodic_memory)

    # -- checkpoints -------------------------------------------------------------------------
    def save(self, filename):
        """One pickle of ``(memory, episodic_memory)`` (reference :60-62); the shared transition
        dicts are pickled once thanks to pickle's memo.  Device back-end: the same structure with
        observations read back from HBM (one dict per transition, one array per distinct
        observation), which the reference's ``load`` reads."""
        if self.store is None:
            with open(filename, "wb") as f:
                pickle.dump((self.memory, self.episodic_memory), f)
            return
        import numpy as np

        obs_memo = {}

        def array_of(obs):
            if obs is None or isinstance(obs, np.ndarray):
                return obs
            key = (tuple(int(r) for r in obs.refs), int(obs.min_seq))
            if key not in obs_memo:
                obs_memo[key] = np.asarray(obs)
            return obs_memo[key]

        flat, episodes = RandomAccessQueue(), RandomAccessQueue()
        for ref in self.episodic_memory:
            ep = []
            for i in range(ref.length):
                d = dict(self.store.entry_view(ref.first + i)[0])
                d["state"], d["next_state"] = array_of(d["state"]), array_of(d["next_state"])
                ep.append(d)
                flat.append([d])
            episodes.append(ep)
        with open(filename, "wb") as f:
            pickle.dump((flat, episodes), f)

    def load(self, filename):
        with open(filename, "rb") as f:
            loaded = pickle.load(f)
        if self.store is not None:
            # re-ingest episode by episode: the transitions go back into HBM
            if isinstance(loaded, tuple):
                episodes = list(loaded[1])
            else:
                episodes, run = [], []
                for item in loaded:
                    tr = item[0] if isinstance(item, list) else item
                    run.append(tr)
                    if tr["is_state_terminal"]:
                        episodes.append(run)
                        run = []
            for ep in episodes:
                tids = []
                for t in ep:
                    extra = {k: v for k, v in t.items() if k not in (
                        "state", "action", "reward", "next_state", "is_state_terminal")
                        and v is not None}
                    self._guard_ring()
                    tids.append(self.store.add_transition(t["state"], t["action"], t["reward"],
                                                          t["next_state"], t["is_state_terminal"],
                                                          extra or None))
                self._commit_device(tids)
            self.store.flush()
            return
        if isinstance(loaded, tuple):
            self.memory, self.episodic_memory = loaded
            return
        # pre-episodic format: a flat sequence of single-transition items.  Episodes are cut at
        # terminal transitions; a trailing unterminated run is not an episode (reference :69-82).
        self.memory = RandomAccessQueue(loaded)
        self.episodic_memory = RandomAccessQueue()
        run = []
        for item in self.memory:
            run.append(item)
            if item["is_state_terminal"]:
                self.episodic_memory.append(run)
                run = []


class _EpisodeRef:
    """An episode of the device back-end: ``length`` consecutive entries from seq ``first``."""

    __slots__ = ("first", "length")

    def __init__(self, first, length):
        self.first, self.length = first, length

    def __len__(self):
        return self.length
