"""HBM-resident payload of the replay buffers.

Layout (all rings, slot = sequence number % ring size):

  frames      DeviceFrameStore     u8 / f32 [F][frame]      one write per frame
  transitions SoA                  state_ref/next_ref int32 [R][k], action
                                   int64 [R] | f32 [R][A], reward f64 [R],
                                   terminal u8 [R]
  entries     n-step windows       e_tids int32 [E][n] (-1 padded), e_len [E]

The host keeps a mirror of the small integer / scalar columns (a few tens of
bytes per transition) for bookkeeping, liveness checks, API-compatible views
and checkpoints; observation bytes exist only in HBM.
"""
import collections

import numpy as np
import torch

from pfrl_amd import ops
from pfrl_amd.device_store import DeviceFrameStore, DeviceObs, recognise_phi
from pfrl_amd.staging import StagingRing, on_stream

_STAGE_ROWS = 4096


class _LazyFramesLike:
    """Duck type of pfrl.wrappers.atari_wrappers.LazyFrames: ``_frames`` list."""


def _frames_of(obs):
    fr = getattr(obs, "_frames", None)
    if isinstance(fr, (list, tuple)) and len(fr) > 0:
        return fr
    return None


class DeviceReplayStore:
    MANY_SETS = int(__import__("os").environ.get("PFRL_MANY_SETS", "2"))

    def __init__(self, device, capacity, num_steps, max_size=None, slack=None, frame_slots=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceReplayStore needs a CUDA/HIP device")
        from pfrl_amd import _native

        _native.lib()  # fail loudly when the HIP library is missing
        self.capacity = capacity
        self.n = int(num_steps)
        bound = capacity if capacity is not None else (max_size or (1 << 20))
        self.bound = int(bound)
        slack = int(slack) if slack is not None else max(65536, 64 * self.n)
        self.slack = slack
        self.E = self.bound + slack
        self.R = self.bound + slack
        self.frame_slots = frame_slots
        self.frames = None            # DeviceFrameStore
        self._own_frames = False
        self.k = None
        self.act_dim = None
        self.desc = None
        self.n_trans = 0              # transitions appended so far (tid counter)
        self.n_entries = 0            # entries appended so far (seq counter)
        self._stage = StagingRing(self.device, slot_bytes=1 << 19, n_slots=32)
        self._frame_stage = None
        self._pend_rows = 0
        self._pend_entries = 0
        self._phi = None
        self._divisor = None
        self._phi_at_ingest = False
        # host ingestion caches (identity de-duplication)
        self._obs_cache = collections.OrderedDict()
        self._frame_cache = collections.OrderedDict()
        self._pend_frames = []
        self._pend_frame_slots = []
        self._out_cache = {}
        self._many_views = {}
        self.many_parity = 0
        self.h_action_stale = False
        # optional replay stream (DQN + PER pipelining, see set_side_stream)
        self.side_stream = None
        self.double_buffer = False
        self._out_parity = 0
        self.ready_event = None
        self._events = []
        # host mirrors, allocated with the tables
        self.h_state_ref = self.h_next_ref = self.h_action = None
        self.h_reward = self.h_terminal = self.h_min_fseq = None
        self.h_e_tids = self.h_e_len = self.h_e_min_fseq = self.h_extra = None

    # -- configuration --------------------------------------------------------
    def set_side_stream(self, stream):
        """Run every launch of this store (table / entry appends, frame uploads,
        the fused gather) on ``stream`` instead of the caller's stream, and
        alternate between two sets of minibatch buffers, so that the gather of
        update u+1 can overlap the backward pass of update u.  After each
        ``fetch`` :attr:`ready_event` marks the gather on that stream."""
        self.side_stream = stream
        self.double_buffer = stream is not None
        self._events = [torch.cuda.Event() for _ in range(4)] if stream is not None else []

    def set_phi(self, phi):
        if phi is not self._phi:
            self._phi = phi
            self._divisor = None

    def _alloc_tables(self, k, action):
        dev = self.device
        self.k = int(k)
        a = np.asarray(action)
        if a.dtype.kind in "iub" and a.ndim == 0:
            self.act_dim = 0
            self.t_action = torch.zeros(self.R, dtype=torch.int64, device=dev)
            self.h_action = np.zeros(self.R, dtype=np.int64)
            self._s_action = np.zeros(_STAGE_ROWS, dtype=np.int64)
        else:
            self.act_dim = int(a.size)
            self.t_action = torch.zeros((self.R, self.act_dim), dtype=torch.float32, device=dev)
            self.h_action = np.zeros((self.R, self.act_dim), dtype=np.float32)
            self._s_action = np.zeros((_STAGE_ROWS, self.act_dim), dtype=np.float32)
        self.t_state_ref = torch.zeros((self.R, self.k), dtype=torch.int32, device=dev)
        self.t_next_ref = torch.zeros((self.R, self.k), dtype=torch.int32, device=dev)
        self.t_reward = torch.zeros(self.R, dtype=torch.float64, device=dev)
        self.t_terminal = torch.zeros(self.R, dtype=torch.uint8, device=dev)
        self.e_tids = torch.full((self.E, self.n), -1, dtype=torch.int32, device=dev)
        self.e_len = torch.zeros(self.E, dtype=torch.int32, device=dev)
        self.desc = ops.make_table_desc(self.t_state_ref, self.t_next_ref, self.t_action,
                                        self.t_reward, self.t_terminal, self.e_tids, self.e_len,
                                        self.k, self.n, self.act_dim)
        self.h_state_ref = np.zeros((self.R, self.k), dtype=np.int32)
        self.h_next_ref = np.zeros((self.R, self.k), dtype=np.int32)
        self.h_reward = np.zeros(self.R, dtype=np.float64)
        self.h_terminal = np.zeros(self.R, dtype=np.uint8)
        self.h_min_fseq = np.zeros(self.R, dtype=np.int64)
        self.h_e_tids = -np.ones((self.E, self.n), dtype=np.int64)   # absolute tids
        self.h_e_len = np.zeros(self.E, dtype=np.int32)
        self.h_e_min_fseq = np.zeros(self.E, dtype=np.int64)
        self.h_extra = {}
        # staging
        self._s_slot = np.zeros(_STAGE_ROWS, dtype=np.int32)
        self._s_state = np.zeros((_STAGE_ROWS, self.k), dtype=np.int32)
        self._s_next = np.zeros((_STAGE_ROWS, self.k), dtype=np.int32)
        self._s_reward = np.zeros(_STAGE_ROWS, dtype=np.float64)
        self._s_term = np.zeros(_STAGE_ROWS, dtype=np.uint8)
        self._s_eslot = np.zeros(_STAGE_ROWS, dtype=np.int32)
        self._s_etids = -np.ones((_STAGE_ROWS, self.n), dtype=np.int32)
        self._s_elen = np.zeros(_STAGE_ROWS, dtype=np.int32)

    # -- observation ingestion ------------------------------------------------
    def _adopt_store(self, store):
        if self.frames is None:
            self.frames = store
        elif self.frames is not store:
            raise ValueError("all observations of one replay buffer must share a frame store")

    def _make_own_store(self, frame, k):
        frame = np.asarray(frame)
        dtype = torch.uint8 if frame.dtype == np.uint8 else torch.float32
        slots = self.frame_slots or (self.R + self.R // 8 + 4096)
        self.frames = DeviceFrameStore(slots, frame.shape, dtype, self.device, stack=k)
        self._own_frames = True
        fb = self.frames.frame_bytes
        rows = max(1, min(4096, (8 << 20) // fb))
        self._frame_stage = StagingRing(self.device, slot_bytes=rows * fb + 64 + rows * 4 + 64,
                                        n_slots=4)
        self._frame_stage_rows = rows

    def _ingest_frame(self, frame):
        """Host frame -> (seq, slot); identical array objects are stored once."""
        key = id(frame)
        hit = self._frame_cache.get(key)
        if hit is not None and hit[0] is frame:
            return hit[1], hit[2]
        arr = np.asarray(frame)
        if self.frames.dtype == torch.float32 and arr.dtype != np.float32:
            arr = arr.astype(np.float32)
        seqs, slots = self.frames.alloc(1)
        self._pend_frames.append(np.ascontiguousarray(arr).reshape(self.frames.frame_shape))
        self._pend_frame_slots.append(int(slots[0]))
        self._frame_cache[key] = (frame, int(seqs[0]), int(slots[0]))
        if len(self._frame_cache) > 4096:
            self._frame_cache.popitem(last=False)
        if len(self._pend_frames) >= self._frame_stage_rows:
            self._flush_frames()
        return int(seqs[0]), int(slots[0])

    def ingest(self, obs):
        """Any observation -> (refs int32[k], min_seq)."""
        if isinstance(obs, DeviceObs):
            if self.frames is None or self.frames is obs.store:
                self._adopt_store(obs.store)
                return obs.refs, obs.min_seq
            return self._ingest_foreign(obs)
        key = id(obs)
        hit = self._obs_cache.get(key)
        if hit is not None and hit[0] is obs:
            return hit[1], hit[2]
        if isinstance(obs, tuple):
            raise TypeError("tuple observations are not supported by the device replay store")
        frs = _frames_of(obs)
        if frs is not None:
            # LazyFrames-like: frames are shared between consecutive observations
            if self.frames is None:
                self._make_own_store(frs[0], len(frs))
            pairs = [self._ingest_frame(f) for f in frs]
        else:
            if self._phi_at_ingest or self._needs_phi_at_ingest(obs):
                # arbitrary (pure) phi: evaluate it once, store phi(obs) as f32
                self._phi_at_ingest = True
                arr = np.asarray(self._phi(obs), dtype=np.float32)
            else:
                arr = np.asarray(obs)
                if arr.dtype not in (np.uint8, np.float32):
                    arr = arr.astype(np.float32)
            if self.frames is None:
                if arr.nbytes % 4:
                    raise ValueError("observation size must be a multiple of 4 bytes")
                self._make_own_store(arr, 1)
            pairs = [self._ingest_frame(arr)]
        refs = np.array([p[1] for p in pairs], dtype=np.int32)
        min_seq = min(p[0] for p in pairs)
        self._obs_cache[key] = (obs, refs, min_seq)
        if len(self._obs_cache) > 4096:
            self._obs_cache.popitem(last=False)
        return refs, min_seq

    def ingest_many(self, obs_list):
        """A whole batch of LazyFrames-like host observations (``_frames`` lists; consecutive
        observations of an env share k - 1 frame objects, pfrl/wrappers/vector_frame_stack.py:
        93-105) -> (refs int32 [N, k], min_seq int64 [N]) in ONE pass: frame identity is
        resolved through the same cache as :meth:`ingest`, every frame not seen before gets a
        ring slot from one allocation, and the new frames go up in one stacked transfer.
        Returns None if the batch is not of that form (callers fall back to per-observation
        ``ingest``)."""
        n = len(obs_list)
        if n == 0:
            return None
        first = _frames_of(obs_list[0])
        if first is None:
            return None
        k = len(first)
        if self.frames is None:
            self._make_own_store(first[0], k)
        if not self._own_frames or self.frames.stack != k:
            return None
        for obs in obs_list:
            frs = getattr(obs, "_frames", None)
            if not isinstance(frs, (list, tuple)) or len(frs) != k:
                return None
        refs = np.empty((n, k), dtype=np.int32)
        seqs = np.empty((n, k), dtype=np.int64)
        fcache, ocache = self._frame_cache, self._obs_cache
        new_frames, new_pos = [], []
        for i, obs in enumerate(obs_list):
            hit = ocache.get(id(obs))
            if hit is not None and hit[0] is obs:
                refs[i] = hit[1]
                seqs[i] = hit[2]
                continue
            for j, f in enumerate(obs._frames):
                h = fcache.get(id(f))
                if h is not None and h[0] is f:
                    seqs[i, j] = h[1]
                    refs[i, j] = h[2]
                else:
                    # first sighting in this pass too: later duplicates find the entry
                    fcache[id(f)] = (f, -1, len(new_frames))
                    new_frames.append(f)
                    new_pos.append((i, j))
                    seqs[i, j] = -1
                    refs[i, j] = len(new_frames) - 1
        if new_frames:
            self._flush_frames()
            aseq, aslot = self.frames.alloc(len(new_frames))
            shape = self.frames.frame_shape
            want = np.uint8 if self.frames.dtype == torch.uint8 else np.float32
            block = np.empty((len(new_frames),) + shape, dtype=want)
            for t, f in enumerate(new_frames):
                block[t] = np.asarray(f).reshape(shape)
                fcache[id(f)] = (f, int(aseq[t]), int(aslot[t]))
            # entries that pointed at "the t-th new frame" (seq -1) get their slot now
            pend = seqs < 0
            idx = refs[pend]
            refs[pend] = aslot[idx]
            seqs[pend] = aseq[idx]
            rows = self._frame_stage_rows
            with on_stream(self.side_stream):
                for a in range(0, len(new_frames), rows):
                    src, sl = self._frame_stage.upload([block[a:a + rows], aslot[a:a + rows]])
                    self.frames.write(src.view(self.frames.dtype).view((-1,) + shape), sl)
            while len(fcache) > 8192:
                fcache.popitem(last=False)
        min_seq = seqs.min(axis=1)
        for i, obs in enumerate(obs_list):
            ocache[id(obs)] = (obs, refs[i].copy(), int(min_seq[i]))
        while len(ocache) > 4096:
            ocache.popitem(last=False)
        return refs, min_seq

    def ingest_vectors(self, obs_list):
        """:meth:`ingest` for a list of plain array observations (one frame each: the f32[376]
        MuJoCo-shaped vectors of SAC / TD3 / DDPG) in ONE pass -> (refs int32 [n, 1], min_seq
        int64 [n]): identity resolved through the same cache, every observation not seen before
        gets its ring slot from one allocation, and the new ones go up in one stacked transfer.
        None (nothing touched) when the list is not of that form or the store is not this
        buffer's own one-frame store yet -- callers fall back to per-observation ``ingest``."""
        fr = self.frames
        if (fr is None or not self._own_frames or fr.stack != 1 or self._phi_at_ingest
                or (self._phi is not None and self._divisor is None)):
            return None
        want = np.uint8 if fr.dtype == torch.uint8 else np.float32
        shape = fr.frame_shape
        size = int(np.prod(shape))
        n = len(obs_list)
        refs = np.empty((n, 1), dtype=np.int32)
        seqs = np.empty(n, dtype=np.int64)
        ocache = self._obs_cache
        new, pos, first = [], [], {}
        for i, obs in enumerate(obs_list):
            hit = ocache.get(id(obs))
            if hit is not None and hit[0] is obs:
                refs[i, 0] = hit[1][0]
                seqs[i] = hit[2]
                continue
            if not (type(obs) is np.ndarray and obs.size == size
                    and (obs.dtype == want or (want is np.float32 and obs.dtype.kind in "fiub"))):
                return None
            t = first.get(id(obs))
            if t is None:
                t = first[id(obs)] = len(new)
                new.append(obs)
            pos.append((i, t))
        if new:
            self._flush_frames()
            aseq, aslot = fr.alloc(len(new))
            block = np.empty((len(new),) + tuple(shape), dtype=want)
            for t, obs in enumerate(new):
                block[t] = obs.reshape(shape)          # (the f32 cast of ingest(), where needed)
            rows = self._frame_stage_rows
            with on_stream(self.side_stream):
                for a in range(0, len(new), rows):
                    src, sl = self._frame_stage.upload([block[a:a + rows],
                                                        np.asarray(aslot[a:a + rows], dtype=np.int32)])
                    fr.write(src.view(fr.dtype).view((-1,) + tuple(shape)), sl)
            for i, t in pos:
                refs[i, 0] = aslot[t]
                seqs[i] = aseq[t]
            for t, obs in enumerate(new):
                ocache[id(obs)] = (obs, np.array([aslot[t]], dtype=np.int32), int(aseq[t]))
            while len(ocache) > 4096:
                ocache.popitem(last=False)
        return refs, seqs

    def _ingest_foreign(self, obs):
        """A device observation whose frames live in ANOTHER frame store (a device env that
        keeps feeding a buffer which already owns a store, e.g. after ``load()`` of a
        reference-format checkpoint): each frame is copied device-to-device into this
        buffer's ring once (consecutive observations share k - 1 frames)."""
        src = obs.store
        self._flush_frames()          # keep ring writes in allocation order
        last = src.next_seq - 1
        seqs_there = [last - ((last - int(r)) % src.n_slots) for r in obs.refs]   # latest writes
        own = self.frames
        if src.dtype != own.dtype:
            raise ValueError("observation dtype %s does not fit this buffer's frame store (%s)"
                             % (src.dtype, own.dtype))
        if src.frame_shape == own.frame_shape and self.k == len(obs.refs):
            pairs = []
            for slot, seq_there in zip((int(r) for r in obs.refs), seqs_there):
                key = ("dev", id(src), seq_there)
                hit = self._frame_cache.get(key)
                if hit is None:
                    seqs, slots = own.alloc(1)
                    own.frames[int(slots[0])].copy_(src.frames[slot])
                    hit = self._frame_cache[key] = (src, int(seqs[0]), int(slots[0]))
                    if len(self._frame_cache) > 4096:
                        self._frame_cache.popitem(last=False)
                pairs.append((hit[1], hit[2]))
            return (np.array([p[1] for p in pairs], dtype=np.int32), min(p[0] for p in pairs))
        if self.k == 1 and int(np.prod(own.frame_shape)) == len(obs.refs) * int(np.prod(src.frame_shape)):
            # this buffer stores whole observations (it was filled from materialised arrays,
            # e.g. a reference-format checkpoint): stack the k frames into one of its frames
            key = ("devobs", id(src)) + tuple(seqs_there)
            hit = self._frame_cache.get(key)
            if hit is None:
                seqs, slots = own.alloc(1)
                idx = torch.as_tensor(np.asarray(obs.refs, dtype=np.int64), device=self.device)
                own.frames[int(slots[0])].view(-1).copy_(src.frames.index_select(0, idx).view(-1))
                hit = self._frame_cache[key] = (src, int(seqs[0]), int(slots[0]))
                if len(self._frame_cache) > 4096:
                    self._frame_cache.popitem(last=False)
            return np.array([hit[2]], dtype=np.int32), hit[1]
        raise ValueError("observation frames %s x %d do not fit this buffer's frame store %s x %d"
                         % (src.frame_shape, len(obs.refs), own.frame_shape, self.k))

    def _needs_phi_at_ingest(self, obs):
        """Arbitrary phi: apply it once on the host when the observation enters
        the buffer and store phi(obs) as f32 (phi must be a pure function)."""
        if self._phi is None or self._divisor is not None:
            return False
        d = recognise_phi(self._phi, obs)
        if d is not None:
            self._divisor = d
            return False
        return True

    def _flush_frames(self):
        n = len(self._pend_frames)
        if n == 0:
            return
        block = np.stack(self._pend_frames)
        slots = np.asarray(self._pend_frame_slots, dtype=np.int32)
        src, sl = self._frame_stage.upload([block, slots])
        self.frames.write(src.view(self.frames.dtype).view((n,) + self.frames.frame_shape), sl)
        self._pend_frames, self._pend_frame_slots = [], []

    # -- transitions / entries ------------------------------------------------
    def add_transition(self, state, action, reward, next_state, terminal, extra=None):
        s_refs, s_seq = self.ingest(state)
        if next_state is None:
            n_refs, n_seq = s_refs, s_seq
        else:
            n_refs, n_seq = self.ingest(next_state)
        if self.desc is None:
            self._alloc_tables(len(s_refs), action)
        if self._pend_rows == _STAGE_ROWS:
            self.flush()
        tid = self.n_trans
        slot = tid % self.R
        i = self._pend_rows
        self._s_slot[i] = slot
        self._s_state[i] = s_refs
        self._s_next[i] = n_refs
        self._s_action[i] = action
        self._s_reward[i] = reward
        self._s_term[i] = terminal
        self._pend_rows = i + 1
        self.h_state_ref[slot] = s_refs
        self.h_next_ref[slot] = n_refs
        self.h_action[slot] = action
        self.h_reward[slot] = reward
        self.h_terminal[slot] = terminal
        self.h_min_fseq[slot] = s_seq if s_seq < n_seq else n_seq
        if self.h_extra:
            # the row this append overwrites (tid - R) takes its extras with it: the tables stay
            # bounded by the ring, like every other column (they used to grow with total steps)
            dead = tid - self.R
            for table in self.h_extra.values():
                table.pop(dead, None)
        if extra:
            for key, val in extra.items():
                self.h_extra.setdefault(key, {})[tid] = val
        self.n_trans = tid + 1
        return tid

    def add_transitions_n1(self, s_refs, n_refs, actions, rewards, terminals, min_seq):
        """``m`` transitions AND their one-transition entries at once (num_steps == 1: every
        append emits exactly the window [tid], pfrl/replay_buffers/replay_buffer.py:53-62):
        the same rows, mirrors and counters as ``m`` calls of add_transition + add_entry,
        written with array operations.  Observations are frame-slot refs of this buffer's own
        frame store.  Returns the first new entry seq."""
        m = len(rewards)
        assert self.n == 1 and self.desc is not None and self.act_dim == 0
        first_seq = self.n_entries
        done = 0
        while done < m:
            room = _STAGE_ROWS - max(self._pend_rows, self._pend_entries)
            if room == 0:
                self.flush()
                continue
            c = min(room, m - done)
            sl = slice(done, done + c)
            tid = np.arange(self.n_trans, self.n_trans + c, dtype=np.int64)
            tslot = tid % self.R
            i = self._pend_rows
            self._s_slot[i:i + c] = tslot
            self._s_state[i:i + c] = s_refs[sl]
            self._s_next[i:i + c] = n_refs[sl]
            self._s_action[i:i + c] = actions[sl]
            self._s_reward[i:i + c] = rewards[sl]
            self._s_term[i:i + c] = terminals[sl]
            self._pend_rows = i + c
            self.h_state_ref[tslot] = s_refs[sl]
            self.h_next_ref[tslot] = n_refs[sl]
            self.h_action[tslot] = actions[sl]
            self.h_reward[tslot] = rewards[sl]
            self.h_terminal[tslot] = terminals[sl]
            self.h_min_fseq[tslot] = min_seq[sl]
            self.n_trans += c
            seq = np.arange(self.n_entries, self.n_entries + c, dtype=np.int64)
            eslot = seq % self.E
            j = self._pend_entries
            self._s_eslot[j:j + c] = eslot
            self._s_etids[j:j + c, 0] = tslot
            self._s_elen[j:j + c] = 1
            self.h_e_tids[eslot, 0] = tid
            self.h_e_len[eslot] = 1
            self.h_e_min_fseq[eslot] = min_seq[sl]
            self._pend_entries = j + c
            self.n_entries += c
            done += c
        return first_seq

    def add_entry(self, tids, span_check=True):
        """An emitted n-step window (list of absolute tids) -> entry seq.  ``span_check=False``:
        the caller (episodic back-ends, whose windows are single transitions of episodes that
        interleave with every other env's) guards the ring itself."""
        first = tids[0]
        if span_check and self.n_trans - first > self.slack:
            raise RuntimeError(
                "replay transition ring too small: an n-step window spans %d transitions; "
                "increase `slack`" % (self.n_trans - first))
        if self._pend_entries == _STAGE_ROWS:
            self.flush()
        seq = self.n_entries
        slot = seq % self.E
        i = self._pend_entries
        ln = len(tids)
        self._s_eslot[i] = slot
        row = self._s_etids[i]
        row[:] = -1
        hrow = self.h_e_tids[slot]
        hrow[:] = -1
        mf = None
        for j, t in enumerate(tids):
            ts = t % self.R
            row[j] = ts
            hrow[j] = t
            f = self.h_min_fseq[ts]
            mf = f if mf is None or f < mf else mf
        self._s_elen[i] = ln
        self.h_e_len[slot] = ln
        self.h_e_min_fseq[slot] = mf
        self._pend_entries = i + 1
        self.n_entries = seq + 1
        return seq

    def take_pending(self):
        """Staged transition rows and entries as arrays for a SHARED staging transfer, and the
        launches to make once they are on the device (see PrioritizedBuffer.take_pending); None
        when there is nothing to ship or host frames are waiting (those go through flush())."""
        if self._pend_frames or not (self._pend_rows or self._pend_entries):
            return None
        r, e = self._pend_rows, self._pend_entries
        arrays = []
        if r:
            arrays += [self._s_slot[:r].copy(), self._s_state[:r].copy(), self._s_next[:r].copy(),
                       self._s_action[:r].copy(), self._s_reward[:r].copy(), self._s_term[:r].copy()]
        if e:
            arrays += [self._s_eslot[:e].copy(), self._s_etids[:e].copy(), self._s_elen[:e].copy()]
        self._pend_rows = self._pend_entries = 0
        desc = self.desc

        def launch(*views):
            k = 0
            if r:
                ops.table_append(desc, *views[:6])
                k = 6
            if e:
                ops.entries_append(desc, *views[k:k + 3])

        return arrays, launch

    def flush(self):
        """Ship staged frames, transition rows and entries to HBM (async)."""
        if not (self._pend_frames or self._pend_rows or self._pend_entries):
            return
        with on_stream(self.side_stream):
            if self._pend_frames:
                self._flush_frames()
            r = self._pend_rows
            if r:
                up = self._stage.upload([self._s_slot[:r], self._s_state[:r], self._s_next[:r],
                                         self._s_action[:r], self._s_reward[:r],
                                         self._s_term[:r]])
                ops.table_append(self.desc, *up)
                self._pend_rows = 0
            e = self._pend_entries
            if e:
                up = self._stage.upload([self._s_eslot[:e], self._s_etids[:e], self._s_elen[:e]])
                ops.entries_append(self.desc, *up)
                self._pend_entries = 0

    # -- sampling -------------------------------------------------------------
    def slots_for(self, seqs):
        """Entry seqs (numpy int64) -> device int32 slots, with liveness check."""
        slots = (seqs % self.E).astype(np.int32)
        if self.frames is not None:
            oldest = self.h_e_min_fseq[slots].min() if len(slots) else 0
            if oldest < self.frames.oldest_live_seq():
                raise RuntimeError(
                    "frame ring too small: a sampled transition references frame %d but the "
                    "ring (n_slots=%d) has already wrapped past it; allocate the "
                    "DeviceFrameStore with at least capacity + num_envs * (stack + num_steps + 2)"
                    " slots" % (oldest, self.frames.n_slots))
        with on_stream(self.side_stream):
            (slots_dev,) = self._stage.upload([slots])
        return slots_dev

    def divisor_for(self, phi):
        if self._phi_at_ingest:
            return 1.0
        if phi is not self._phi or self._divisor is None:
            self._phi = phi
            if self.frames.dtype == torch.float32:
                sample = self.frames.frames[:1].cpu().numpy()[0]
            else:
                sample = self.frames.frames[: self.k].cpu().numpy()
                if sample.shape[0] == 1:
                    sample = sample[0]
                elif sample.ndim >= 3 and sample.shape[1] == 1:
                    sample = np.concatenate(list(sample), axis=0)
                if not np.any(sample):
                    sample = sample.copy()
                    sample.reshape(-1)[: 256] = np.arange(min(256, sample.size), dtype=np.uint8)
            d = recognise_phi(phi, sample)
            if d is None:
                raise TypeError(
                    "pfrl_amd: phi is not a cast/scale feature extractor and the observations "
                    "are already device-resident; use phi(x) = float32(x) / c")
            self._divisor = d
        return self._divisor

    def _out_buffers(self, B, tag):
        """Persistent fp32 minibatch buffers (stable addresses let the update
        be replayed from a captured HIP graph)."""
        nhwc = bool(self.frames.emit_channels_last
                    and ops.channels_last_supported(self.frames.frames, self.k))
        key = (tag, B, nhwc)
        out = self._out_cache.get(key)
        if out is not None:
            return out
        dev = self.device
        fshape = self.frames.frame_shape
        k = self.k
        if k == 1:
            oshape = (B,) + fshape
        elif len(fshape) >= 2 and fshape[0] == 1:
            oshape = (B, k) + fshape[1:]        # LazyFrames: concatenate on axis 0
        else:
            oshape = (B, k) + fshape
        if nhwc:
            hw = oshape[-2:]
            new_obs = lambda: ops.empty_channels_last(B, hw, dev)
        else:
            new_obs = lambda: torch.empty(oshape, dtype=torch.float32, device=dev)
        out = dict(
            state=new_obs(),
            next_state=new_obs(),
            action=(torch.empty(B, dtype=torch.int64, device=dev) if self.act_dim == 0 else
                    torch.empty((B, self.act_dim), dtype=torch.float32, device=dev)),
            reward=torch.empty(B, dtype=torch.float32, device=dev),
            is_state_terminal=torch.empty(B, dtype=torch.float32, device=dev),
            discount=torch.empty(B, dtype=torch.float32, device=dev),
        )
        self._out_cache[key] = out
        return out

    def fetch(self, batch, phi, gamma):
        """The fused batch_experiences launch for a DeviceExperienceBatch."""
        self.flush()
        B = len(batch)
        tag = "single"
        if self.double_buffer:
            self._out_parity ^= 1
            tag = "single%d" % self._out_parity
        out = dict(self._out_buffers(B, tag))
        gp = [gamma ** i for i in range(self.n + 1)]
        divisor = self.divisor_for(phi)
        with on_stream(self.side_stream):
            ops.batch_experiences(self.desc, self.frames.frames, divisor, batch.slots_dev, gp, out)
            if self.side_stream is not None:
                ev = self._events[0]
                self._events = self._events[1:] + [ev]
                ev.record()
                self.ready_event = ev
        if batch.weights_dev is not None:
            out["weights"] = batch.weights_dev
        return out

    def fetch_many(self, seq_sets, phi, gamma):
        """All minibatches of one env step in ONE launch (U * B entries)."""
        self.flush()
        U = len(seq_sets)
        B = len(seq_sets[0])
        seqs = np.concatenate(seq_sets)
        slots_dev = self.slots_for(seqs)
        flat = self._out_buffers(U * B, "many")
        gp = [gamma ** i for i in range(self.n + 1)]
        with on_stream(self.side_stream):
            ops.batch_experiences(self.desc, self.frames.frames, self.divisor_for(phi), slots_dev,
                                  gp, flat)
        return {k: v.view((U, B) + tuple(v.shape[1:])) for k, v in flat.items()}

    def fetch_many_slots(self, slots_dev, U, B, phi, gamma, alternate=True):
        """``fetch_many`` for entry slots that are already on the device (planned natively,
        liveness checked by the planner): the one fused gather, [U, B, ...] views.
        ``alternate=False``: always the buffers of :meth:`fetch_many` (callers whose host
        cannot run a step ahead anyway -- a host env waits for the actions -- keep one graph
        per range instead of MANY_SETS)."""
        self.flush()
        if not alternate:
            flat = self._out_buffers(U * B, "many")
            gp = [gamma ** i for i in range(self.n + 1)]
            with on_stream(self.side_stream):
                ops.batch_experiences(self.desc, self.frames.frames, self.divisor_for(phi),
                                      slots_dev, gp, flat)
            return {k: v.view((U, B) + tuple(v.shape[1:])) for k, v in flat.items()}
        # Alternate between MANY_SETS sets of minibatch buffers: every set has its own captured
        # range graph, and launching a graph while ITS previous replay is still running makes
        # hipGraphLaunch wait on the host -- with one set the host could never be more than one
        # step ahead of the GPU (measured: 0.45 ms of host-paced gaps at every step boundary).
        self.many_parity = (self.many_parity + 1) % self.MANY_SETS
        flat = self._out_buffers(U * B, "many%d" % self.many_parity)
        gp = [gamma ** i for i in range(self.n + 1)]
        with on_stream(self.side_stream):
            ops.batch_experiences(self.desc, self.frames.frames, self.divisor_for(phi), slots_dev,
                                  gp, flat)
        key = (U, B, id(flat))
        views = self._many_views.get(key)
        if views is None or views[0] is not flat:
            if len(self._many_views) > 16:
                self._many_views.clear()
            views = self._many_views[key] = (
                flat, {k: v.view((U, B) + tuple(v.shape[1:])) for k, v in flat.items()})
        return dict(views[1])

    def fetch_episodes(self, windows, phi, gamma):
        """Sampled episode windows [(first entry seq, length), ...] sorted by descending length
        -> the dict of ``batch_recurrent_experiences`` (reference pfrl/replay_buffer.py:219-287):
        ``state`` / ``next_state`` as per-episode views of ONE gathered tensor, the scalar
        columns flat in packed (time-major) order.  One launch (pfrl_batch_episodes)."""
        self.flush()
        n = len(windows)
        lens = np.array([w[1] for w in windows], dtype=np.int64)
        assert n > 0 and lens.min() >= 1 and (lens[:-1] >= lens[1:]).all()
        rows, T = int(lens.sum()), int(lens[0])
        ep_first = np.array([w[0] % self.E for w in windows], dtype=np.int32)
        ep_row0 = np.zeros(n + 1, dtype=np.int32)
        ep_row0[1:] = np.cumsum(lens)
        # row_start[t] = packed rows before step t = sum over episodes of min(len, t)
        row_start = np.minimum(lens[None, :], np.arange(T + 1)[:, None]).sum(axis=1).astype(np.int32)
        if self.frames is not None:
            oldest = min(int(self.h_e_min_fseq[(w[0] + np.arange(w[1])) % self.E].min())
                         for w in windows)
            if oldest < self.frames.oldest_live_seq():
                raise RuntimeError("frame ring too small: a sampled episode references frame %d "
                                   "but the ring (n_slots=%d) has wrapped past it"
                                   % (oldest, self.frames.n_slots))
        with on_stream(self.side_stream):
            f_dev, r0_dev, rs_dev = self._stage.upload([ep_first, ep_row0, row_start])
            dev = self.device
            fshape, k = self.frames.frame_shape, self.k
            if k == 1:
                oshape = (rows,) + fshape
            elif len(fshape) >= 2 and fshape[0] == 1:
                oshape = (rows, k) + fshape[1:]
            else:
                oshape = (rows, k) + fshape
            out = dict(
                state=torch.empty(oshape, dtype=torch.float32, device=dev),
                next_state=torch.empty(oshape, dtype=torch.float32, device=dev),
                action=(torch.empty(rows, dtype=torch.int64, device=dev) if self.act_dim == 0 else
                        torch.empty((rows, self.act_dim), dtype=torch.float32, device=dev)),
                reward=torch.empty(rows, dtype=torch.float32, device=dev),
                is_state_terminal=torch.empty(rows, dtype=torch.float32, device=dev),
                discount=torch.empty(rows, dtype=torch.float32, device=dev))
            ops.batch_episodes(self.desc, self.frames.frames, self.divisor_for(phi), f_dev, r0_dev,
                               rs_dev, n, T, rows, self.E, gamma, out)
        bounds = ep_row0.tolist()
        out["state"] = [out["state"][a:b] for a, b in zip(bounds[:-1], bounds[1:])]
        out["next_state"] = [out["next_state"][a:b] for a, b in zip(bounds[:-1], bounds[1:])]
        return out

    def _sync_actions(self):
        """The native append path leaves the action column on the device only; host views
        (entry_view, save) read it back once."""
        if self.h_action_stale:
            self.flush()
            self.h_action[...] = self.t_action.cpu().numpy()
            self.h_action_stale = False

    # -- API-compatible host views ---------------------------------------------
    def transition_view(self, tid, weight=None):
        self._sync_actions()
        slot = tid % self.R
        store = self.frames
        d = dict(
            state=DeviceObs(store, self.h_state_ref[slot].copy(), int(self.h_min_fseq[slot])),
            action=(int(self.h_action[slot]) if self.act_dim == 0 else self.h_action[slot].copy()),
            reward=float(self.h_reward[slot]),
            next_state=DeviceObs(store, self.h_next_ref[slot].copy(), int(self.h_min_fseq[slot])),
            next_action=None,
            is_state_terminal=bool(self.h_terminal[slot]),
        )
        for key, table in self.h_extra.items():
            if tid in table:
                d[key] = table[tid]
        if weight is not None:
            d["weight"] = weight
        return d

    def entry_view(self, seq, weight=None):
        slot = seq % self.E
        ln = int(self.h_e_len[slot])
        tids = [int(t) for t in self.h_e_tids[slot][:ln]]
        return [self.transition_view(t, weight if j == 0 else None) for j, t in enumerate(tids)]


    # -- native checkpoint ------------------------------------
    _TABLES = ("t_state_ref", "t_next_ref", "t_action", "t_reward", "t_terminal", "e_tids", "e_len")
    _MIRRORS = ("h_state_ref", "h_next_ref", "h_action", "h_reward", "h_terminal", "h_min_fseq",
                "h_e_tids", "h_e_len", "h_e_min_fseq")

    def state_dict(self, head_seq):
        """Everything needed to continue sampling: tables, host mirrors, counters
        and the frames still referenced by live entries (only those are read back
        from HBM).  ``head_seq`` is the entry sequence number of logical index 0."""
        self.flush()
        if self.desc is not None:
            self._sync_actions()
        out = dict(version=1, n=self.n, k=self.k, act_dim=self.act_dim, bound=self.bound,
                   slack=self.slack, n_trans=self.n_trans, n_entries=self.n_entries,
                   extra=self.h_extra)
        if self.desc is None:
            return out
        for name in self._TABLES:
            out[name] = getattr(self, name).cpu()
        for name in self._MIRRORS:
            out[name] = getattr(self, name)
        fr = self.frames
        live = np.arange(head_seq, self.n_entries) % self.E
        oldest = int(self.h_e_min_fseq[live].min()) if len(live) else fr.next_seq
        oldest = max(oldest, fr.oldest_live_seq(), 0)
        seqs = np.arange(oldest, fr.next_seq, dtype=np.int64)
        idx = torch.from_numpy(seqs % fr.n_slots).to(self.device)
        out.update(frame_shape=fr.frame_shape, frame_dtype=str(fr.dtype).replace("torch.", ""),
                   frame_slots=fr.n_slots, frame_stack=fr.stack, frame_first_seq=oldest,
                   frame_next_seq=fr.next_seq, frame_data=fr.frames[idx].cpu(),
                   own_frames=self._own_frames, phi_at_ingest=self._phi_at_ingest)
        return out

    def load_state_dict(self, sd):
        assert sd["version"] == 1 and sd["n"] == self.n
        assert sd["bound"] == self.bound and sd["slack"] == self.slack, \
            "checkpoint was written with a different capacity / slack"
        self.n_trans, self.n_entries = sd["n_trans"], sd["n_entries"]
        self.h_extra = sd.get("extra", {})
        if "t_reward" not in sd:
            return
        if self.frames is None:
            dtype = getattr(torch, sd["frame_dtype"])
            self.frames = DeviceFrameStore(sd["frame_slots"], sd["frame_shape"], dtype,
                                           self.device, stack=sd["frame_stack"])
            self._own_frames = True
            fb = self.frames.frame_bytes
            rows = max(1, min(4096, (8 << 20) // fb))
            self._frame_stage = StagingRing(self.device,
                                            slot_bytes=rows * fb + 64 + rows * 4 + 64, n_slots=4)
            self._frame_stage_rows = rows
        fr = self.frames
        assert fr.n_slots == sd["frame_slots"] and fr.frame_shape == tuple(sd["frame_shape"])
        seqs = np.arange(sd["frame_first_seq"], sd["frame_next_seq"], dtype=np.int64)
        if len(seqs):
            idx = torch.from_numpy(seqs % fr.n_slots).to(self.device)
            fr.frames[idx] = sd["frame_data"].to(self.device)
        fr.next_seq = sd["frame_next_seq"]
        self._phi_at_ingest = sd.get("phi_at_ingest", False)
        action_probe = (np.zeros((), dtype=np.int64) if sd["act_dim"] == 0
                        else np.zeros(sd["act_dim"], dtype=np.float32))
        if self.desc is None:
            self._alloc_tables(sd["k"], action_probe)
        for name in self._TABLES:
            getattr(self, name).copy_(sd[name].to(self.device))
        for name in self._MIRRORS:
            getattr(self, name)[...] = sd[name]
