"""Experience replay buffer with n-step windows.

Mirrors ``pfrl.replay_buffers.ReplayBuffer``
(/root/reference/pfrl/replay_buffers/replay_buffer.py:9-94): constructor
``(capacity=None, num_steps=1)``, ``append`` with per-``env_id`` n-step
windows (:33-62), ``stop_current_episode`` (:64-76), ``sample`` (:78-80),
``__len__``, ``save`` / ``load`` and the same assertions.

Storage back-ends
  * device (``device='cuda:N'`` or bound by an agent created with ``gpu>=0``):
    observation bytes, transition columns and n-step entries live in HBM
    (:class:`DeviceReplayStore`); ``sample`` returns a
    :class:`DeviceExperienceBatch` that ``batch_experiences`` turns into fp32
    minibatches with one fused HIP launch.  There is no CPU fallback on this
    path: a missing HIP library raises.
  * host (no device, or an agent created with ``gpu=None/-1``): the plumbing
    path of config 1 -- Python lists of transition dicts, as in the reference.
"""
import collections
import pickle

import numpy as np
import torch

from pfrl_amd import replay_buffer
from pfrl_amd.collections.random_access_queue import RandomAccessQueue
from pfrl_amd.replay_buffer import DeviceExperienceBatch
from pfrl_amd.utils.random import sample_n_k


class _ReferenceUnpickler(pickle.Unpickler):
    """Reads pickles written by the reference itself (``pickle.dump(self.memory)`` names
    ``pfrl.collections.random_access_queue.RandomAccessQueue`` and, for LazyFrames
    observations, ``pfrl.wrappers.atari_wrappers.LazyFrames``) when only this package is
    installed: ``pfrl.x.y`` resolves to ``pfrl_amd.x.y``."""

    def find_class(self, module, name):
        if module == "pfrl" or module.startswith("pfrl."):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return super().find_class("pfrl_amd" + module[len("pfrl"):], name)
        return super().find_class(module, name)


class _DeviceQueue:
    """FIFO view over the entry ring: logical index i <-> entry seq head + i
    (the role RandomAccessQueue plays in the reference)."""

    def __init__(self, store, maxlen):
        self.store = store
        self.maxlen = maxlen
        self.head = 0  # seq of logical index 0

    def __len__(self):
        return self.store.n_entries - self.head

    def append_entry(self, tids):
        self.store.add_entry(tids)
        if self.maxlen is not None and len(self) > self.maxlen:
            self.head += 1
        elif self.maxlen is None and len(self) > self.store.bound:
            raise RuntimeError(
                "unbounded ReplayBuffer exceeded its device allocation (max_size=%d)"
                % self.store.bound)

    def __getitem__(self, i):
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("replay index out of range")
        return self.store.entry_view(self.head + i)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def sample(self, k):
        idx = sample_n_k(len(self), k)
        seqs = self.head + np.asarray(idx, dtype=np.int64)
        return DeviceExperienceBatch(self.store, self.store.slots_for(seqs), seqs)


class ReplayBuffer(replay_buffer.AbstractReplayBuffer):
    """Experience Replay Buffer (uniform sampling, optional N-step)."""

    capacity = None

    def __init__(self, capacity=None, num_steps=1, device=None, max_size=None, slack=None,
                 frame_slots=None):
        self.capacity = capacity
        assert num_steps > 0
        self.num_steps = num_steps
        self._device_opts = dict(max_size=max_size, slack=slack, frame_slots=frame_slots)
        self.device = None
        self.store = None
        self.memory = None
        self.last_n_transitions = collections.defaultdict(
            lambda: collections.deque([], maxlen=num_steps))
        if device is not None:
            self.bind(device)

    # -- back-end selection -----------------------------------------------------
    def _make_memory_host(self):
        return RandomAccessQueue(maxlen=self.capacity)

    def _make_memory_device(self):
        return _DeviceQueue(self.store, self.capacity)

    def bind(self, device, phi=None):
        """Choose the storage back-end.  Called by agents with their device."""
        device = torch.device(device)
        if self.memory is not None:
            if self.device != device:
                raise RuntimeError("replay buffer already bound to %s" % self.device)
            if self.store is not None and phi is not None:
                self.store.set_phi(phi)
            return self
        self.device = device
        if device.type == "cuda":
            from pfrl_amd.replay_buffers.device_replay import DeviceReplayStore

            self.store = DeviceReplayStore(device, self.capacity, self.num_steps,
                                           **self._device_opts)
            if phi is not None:
                self.store.set_phi(phi)
            self.memory = self._make_memory_device()
        else:
            self.memory = self._make_memory_host()
        return self

    def _ensure_bound(self):
        if self.memory is None:
            self.bind(torch.device("cpu"))

    @property
    def is_device(self):
        return self.store is not None

    # -- replay stream (overlap of replay launches with the network update) ---------
    def set_replay_stream(self, stream):
        """Route every device launch of this buffer to ``stream`` (None = back to the
        caller's stream).  The caller owns the ordering: see DQN._update_from_batch."""
        self._ensure_bound()
        if self.store is None:
            raise RuntimeError("replay streams need the device back-end")
        self.store.set_side_stream(stream)
        tree = getattr(self.memory, "tree", None)
        if tree is not None:
            tree.side_stream = stream

    @property
    def replay_stream(self):
        return None if self.store is None else self.store.side_stream

    def replay_stream_wait_current(self):
        """Everything enqueued so far on the caller's stream (e.g. the frames of
        this env step) happens-before later replay launches."""
        st = self.replay_stream
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(self.device))

    def current_wait_replay_stream(self):
        """The caller's stream waits for the replay launches enqueued so far."""
        st = self.replay_stream
        if st is not None:
            torch.cuda.current_stream(self.device).wait_stream(st)

    # -- reference API ------------------------------------------------------------
    def _emit(self, window):
        if self.store is not None:
            self.memory.append_entry(list(window))
        else:
            self.memory.append(list(window))

    def append(self, state, action, reward, next_state=None, next_action=None,
               is_state_terminal=False, env_id=0, **kwargs):
        self._ensure_bound()
        window = self.last_n_transitions[env_id]
        if self.store is not None:
            if next_action is not None:
                kwargs = dict(kwargs, next_action=next_action)
            item = self.store.add_transition(state, action, reward, next_state, is_state_terminal,
                                             kwargs or None)
        else:
            item = dict(state=state, action=action, reward=reward, next_state=next_state,
                        next_action=next_action, is_state_terminal=is_state_terminal, **kwargs)
        window.append(item)
        if is_state_terminal:
            # flush every suffix of the window (reference :55-59)
            while window:
                self._emit(window)
                del window[0]
        elif len(window) == self.num_steps:
            self._emit(window)

    def stop_current_episode(self, env_id=0):
        self._ensure_bound()
        window = self.last_n_transitions[env_id]
        # a full window has already been emitted by append (reference :66-72)
        if 0 < len(window) < self.num_steps:
            self._emit(window)
        if 0 < len(window) <= self.num_steps:
            del window[0]
        while window:
            self._emit(window)
            del window[0]

    def sample(self, num_experiences):
        self._ensure_bound()
        assert len(self.memory) >= num_experiences
        return self.memory.sample(num_experiences)

    def __len__(self):
        return 0 if self.memory is None else len(self.memory)

    # -- step-fused sampling (device, uniform) ----------------------------------
    @property
    def supports_lookahead(self):
        return self.store is not None and type(self.memory) is _DeviceQueue

    def batch_append_supported(self, obs_batch):
        """The array form of ``append`` below applies: uniform device replay with one-step
        entries whose tables exist, discrete actions, and observations that are refs into this
        buffer's own frame store."""
        st = self.store
        return (st is not None and type(self.memory) is _DeviceQueue and self.num_steps == 1
                and st.desc is not None and st.act_dim == 0 and st.frames is not None
                and getattr(obs_batch, "store", None) is st.frames and not st._phi_at_ingest)

    def vector_range_append_supported(self):
        """The native append + draw walk of the vector-observation agents applies
        (agents/_vector_device_step.py): uniform device replay with one-step entries whose
        tables exist, float action rows, this buffer's own one-frame store, no per-transition
        extras."""
        st = self.store
        return (st is not None and type(self.memory) is _DeviceQueue and self.num_steps == 1
                and st.desc is not None and st.act_dim is not None and st.act_dim > 0
                and st.k == 1 and st.frames is not None and st._own_frames
                and not st._phi_at_ingest and not st.h_extra
                and type(self).append is ReplayBuffer.append
                and type(self).stop_current_episode is ReplayBuffer.stop_current_episode)

    def append_batch_n1(self, s_refs, s_min_seq, actions, rewards, n_refs, n_min_seq, terminals):
        """``append(...)`` for m envs in env order, num_steps == 1 (each append emits its own
        one-transition entry at once; reference replay_buffer.py:33-62), as array writes.
        Returns (len(self), head) AFTER EACH of the m appends (int64 arrays): what ``len`` and
        the queue head were at that point of the reference's per-env loop, which is all that
        index draws made between the appends depend on."""
        m = len(rewards)
        st, q = self.store, self.memory
        n0, head0 = st.n_entries, q.head
        st.add_transitions_n1(s_refs, n_refs, actions, rewards, terminals,
                              np.minimum(s_min_seq, n_min_seq))
        total = n0 + 1 + np.arange(m, dtype=np.int64)          # entries appended so far
        if q.maxlen is not None:
            heads = np.maximum(head0, total - q.maxlen)
        else:
            if st.n_entries - head0 > st.bound:
                raise RuntimeError("unbounded ReplayBuffer exceeded its device allocation "
                                   "(max_size=%d)" % st.bound)
            heads = np.full(m, head0, dtype=np.int64)
        q.head = int(heads[-1])
        return total - heads, heads

    def lookahead_sample(self, k):
        """Draw the indices ``sample(k)`` would draw NOW (same NumPy stream use)
        and return the entry sequence numbers, without touching the device."""
        assert len(self.memory) >= k
        idx = sample_n_k(len(self.memory), k)
        return self.memory.head + np.asarray(idx, dtype=np.int64)

    def lookahead_sample_at(self, length, head, k):
        """The draw ``sample(k)`` would make at a point of the per-env loop where
        ``len(self) == length`` and the queue head was ``head`` (see append_batch_n1): same
        NumPy stream use as :meth:`lookahead_sample`, entry sequence numbers returned."""
        assert length >= k
        return int(head) + np.asarray(sample_n_k(int(length), k), dtype=np.int64)

    def fetch_many(self, seq_sets, phi, gamma):
        """One fused batch_experiences launch for several planned minibatches;
        returns a dict of tensors with a leading "update" dimension."""
        return self.store.fetch_many(seq_sets, phi, gamma)

    def save(self, filename, native=False, materialize=False):
        """Reference-compatible pickle of the queue (replay_buffer.py:85-88).

        Host back-end: the stored entries as they are (a plain deque of the entry lists), so
        pickle's memo keeps what the reference's own ``save`` keeps -- one dict per transition
        however many n-step windows hold it, one array per frame however many LazyFrames share
        it.  ``materialize=True`` writes observations as NumPy arrays instead (one per distinct
        observation object): a file without any class of the observation's package, which the
        reference reads with nothing else installed, at 4x the size for frame stacks.

        Device back-end: observations are read back from HBM and always materialised (one array
        per distinct observation); with ``native=True`` a compact torch checkpoint of the HBM
        tables and only the live frames instead (see :meth:`load`, which recognises both)."""
        self._ensure_bound()
        if self.store is not None:
            self.store.flush()
            torch.cuda.synchronize(self.device)
        if native and self.store is not None:
            torch.save(self._native_state(), filename)
            return
        with open(filename, "wb") as f:
            if self.store is None and not isinstance(self.memory, RandomAccessQueue):
                # host PrioritizedBuffer: the object itself, priorities included, as the
                # reference pickles it (replay_buffer.py:85-88)
                pickle.dump(self.memory, f)
            else:
                pickle.dump(self._portable_queue(materialize), f)

    def _portable_queue(self, materialize=False):
        """The queue as a plain ``collections.deque`` of n-step entries (lists of transition
        dicts), which the reference's ``ReplayBuffer.load`` reads as it reads its own v0.2 files
        (pfrl/replay_buffers/replay_buffer.py:89-94 wraps a deque into its RandomAccessQueue).
        Shared objects stay shared: a transition dict / observation is converted once, whatever
        number of entries refer to it."""
        if self.store is None and not materialize:
            return collections.deque(iter(self.memory), maxlen=self.capacity)
        obs_memo, trans_memo = {}, {}

        def array_of(obs):
            if obs is None or isinstance(obs, (np.ndarray, tuple)):
                return obs
            key = (tuple(int(r) for r in obs.refs), int(obs.min_seq)) if hasattr(obs, "refs") \
                else id(obs)
            hit = obs_memo.get(key)
            if hit is None:
                hit = obs_memo[key] = (obs, np.asarray(obs))     # (keeps id() keys alive)
            return hit[1]

        def plain(t):
            d = dict(t)
            for key in ("state", "next_state"):
                d[key] = array_of(d.get(key))
            return d

        out = collections.deque(maxlen=self.capacity)
        if self.store is None:
            for entry in self.memory:
                row = []
                for t in entry:
                    d = trans_memo.get(id(t))
                    if d is None:
                        d = trans_memo[id(t)] = (t, plain(t))
                    row.append(d[1])
                out.append(row)
            return out
        head = self.memory.head
        for i in range(len(self.memory)):
            slot = (head + i) % self.store.E
            ln = int(self.store.h_e_len[slot])
            entry = []
            for tid in (int(t) for t in self.store.h_e_tids[slot][:ln]):
                d = trans_memo.get(tid)
                if d is None:
                    d = trans_memo[tid] = plain(self.store.transition_view(tid))
                entry.append(d)
            out.append(entry)
        return out

    def _native_state(self):
        return dict(kind="pfrl_amd.ReplayBuffer", capacity=self.capacity,
                    num_steps=self.num_steps, head=self.memory.head,
                    windows={k: list(v) for k, v in self.last_n_transitions.items()},
                    store=self.store.state_dict(self.memory.head))

    def _load_native(self, sd):
        assert sd["capacity"] == self.capacity and sd["num_steps"] == self.num_steps
        self.store.load_state_dict(sd["store"])
        self.memory.head = sd["head"]
        self.last_n_transitions.clear()
        for k, v in sd["windows"].items():
            self.last_n_transitions[k].extend(v)

    def load(self, filename):
        if self.store is not None:
            try:
                sd = torch.load(filename, weights_only=False)
            except Exception:
                sd = None
            if isinstance(sd, dict) and str(sd.get("kind", "")).startswith("pfrl_amd."):
                return self._load_native(sd)
        with open(filename, "rb") as f:
            loaded = _ReferenceUnpickler(f).load()
        if isinstance(loaded, collections.deque):
            loaded = RandomAccessQueue(loaded, maxlen=loaded.maxlen)
        if self.store is None:
            self._ensure_bound()
            if self.store is None:
                self.memory = loaded
                return
        for entry in loaded:
            tids = [self.store.add_transition(t["state"], t["action"], t["reward"],
                                              t["next_state"], t["is_state_terminal"])
                    for t in entry]
            self.memory.append_entry(tids)
        # pending frame uploads and table rows reach HBM now: the loaded contents are
        # readable (entry views, DeviceObs.to_numpy) without a sample() in between
        self.store.flush()
