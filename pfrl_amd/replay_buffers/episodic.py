"""Replay of whole episodes, for recurrent models.

Mirrors ``pfrl.replay_buffers.EpisodicReplayBuffer``
(/root/reference/pfrl/replay_buffers/episodic.py:9-98): transitions of the episode in progress
are held per ``env_id`` and become visible -- as one item of ``episodic_memory`` and as
``len(episode)`` single-transition items of ``memory`` -- only when the episode ends (terminal
transition or ``stop_current_episode``).  ``capacity`` bounds the number of TRANSITIONS; whole
episodes are evicted oldest-first until it is respected (:91-97), so ``len(buffer)`` never exceeds
it after a commit.

This is the host container of SURVEY.md 8(f) row 4: episodes are lists of the caller's transition
dicts, held by reference exactly as the reference does, and sampling consumes the global NumPy
stream draw-for-draw (``sample_n_k`` for the episode choice, then one ``randint`` per episode that
is cut to ``max_len``).  The HBM frame ring / fused gather of the flat buffers is not used here:
variable-length episode gathers on the device are the next step of that row.
"""
import collections
import pickle

from pfrl_amd.collections.random_access_queue import RandomAccessQueue
from pfrl_amd.replay_buffer import AbstractEpisodicReplayBuffer, random_subseq


class EpisodicReplayBuffer(AbstractEpisodicReplayBuffer):
    capacity = None

    def __init__(self, capacity=None):
        self.capacity = capacity
        self.current_episode = collections.defaultdict(list)
        self.episodic_memory = RandomAccessQueue()
        self.memory = RandomAccessQueue()

    # -- ingest ------------------------------------------------------------------------------
    def append(self, state, action, reward, next_state=None, next_action=None,
               is_state_terminal=False, env_id=0, **kwargs):
        self.current_episode[env_id].append(dict(
            state=state, action=action, reward=reward, next_state=next_state,
            next_action=next_action, is_state_terminal=is_state_terminal, **kwargs))
        if is_state_terminal:
            self.stop_current_episode(env_id=env_id)

    def stop_current_episode(self, env_id=0):
        episode = self.current_episode[env_id]
        if not episode:
            return
        self.current_episode[env_id] = []
        self._commit(episode)

    def _commit(self, episode):
        self.episodic_memory.append(episode)
        for transition in episode:
            self.memory.append([transition])
        if self.capacity is None:
            return
        while len(self.memory) > self.capacity:
            for _ in self.episodic_memory.popleft():
                self.memory.popleft()

    # -- sampling ----------------------------------------------------------------------------
    def sample(self, n):
        assert len(self.memory) >= n
        return self.memory.sample(n)

    def sample_episodes(self, n_episodes, max_len=None):
        assert len(self.episodic_memory) >= n_episodes
        episodes = self.episodic_memory.sample(n_episodes)
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
        dicts are pickled once thanks to pickle's memo."""
        with open(filename, "wb") as f:
            pickle.dump((self.memory, self.episodic_memory), f)

    def load(self, filename):
        with open(filename, "rb") as f:
            loaded = pickle.load(f)
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
