"""Proximal Policy Optimization on the device rollout path.

Mirrors ``pfrl.agents.ppo.PPO`` (/root/reference/pfrl/agents/ppo.py) for the
non-recurrent case: constructor (:320-346), ``batch_act`` / ``batch_observe``
(:706-807), dataset construction (:110-142, :228-244), GAE (:36-47),
advantage standardisation (:476-478, :494-495), minibatch order (:247-257),
loss (:634-671), statistics (:809-817).

Where the reference keeps a Python list of transition dicts and re-collates
observations for every pass, this implementation keeps a T x N rollout on the
device:

  observations   frame slots in a DeviceFrameStore (one write per frame)
  value pass     gather kernel -> model, in chunks sized for HBM
  GAE            pfrl_gae_scan: one lane per env, reverse scan over T, restart
                 at every fragment end (done / reset / rollout end)
  adv statistics pfrl_adv_stats (f64 accumulate, wavefront shuffle reduce)
  minibatches    pfrl_ppo_minibatch gathers advantages (standardised), old
                 log-probs / values, targets, actions and observation refs for
                 the dataset positions drawn on the host with Python's
                 ``random`` exactly as the reference does

so the only per-update host work is the permutation draw.

Created without a GPU (``gpu=None / -1``) the agent runs the reference's list-of-dicts algorithm
instead (``ppo_host.HostRollouts``), which is also where ``recurrent=True`` lives; the HIP kernels
are never involved there and nothing on the device path falls back to it.
"""
import os
import random
from logging import getLogger

import numpy as np
import torch
import torch.nn.functional as F

from pfrl_amd import agent, ops
from pfrl_amd.agents.dqn import _DeviceRecord, _mean_or_nan
from pfrl_amd.utils.clip_l2_grad_norm import clip_grad_norm_device_
from pfrl_amd.device_store import DeviceObs, DeviceObsBatch
from pfrl_amd.utils.batch_states import batch_states
from pfrl_amd.utils.contexts import evaluating
from pfrl_amd.utils.mode_of_distribution import mode_of_distribution


from pfrl_amd.agents.ppo_host import (  # NOQA,E402  (the reference's module-level helpers)
    _add_advantage_and_value_target_to_episode, _add_advantage_and_value_target_to_episodes,
    _add_log_prob_and_value_to_episodes, _add_log_prob_and_value_to_episodes_recurrent,
    _compute_explained_variance, _limit_sequence_length, _make_dataset, _make_dataset_recurrent,
    _yield_minibatches, _yield_subset_of_sequences_with_fixed_number_of_items)


def _elementwise_clip(x, x_min, x_max):
    return torch.min(torch.max(x, x_min), x_max)


def _yield_minibatch_positions(n, minibatch_size, num_epochs):
    """Dataset positions of successive minibatches.  Same consumption of the
    ``random`` stream as reference :247-257 (random.sample over the dataset)."""
    buf = []
    done = 0
    while done < n * num_epochs:
        while len(buf) < minibatch_size:
            buf = random.sample(range(n), k=n) + buf
        yield buf[-minibatch_size:]
        done += minibatch_size
        buf = buf[:-minibatch_size]


def _random_permutation(n):
    """``random.sample(range(n), k=n)`` as an int64 array: the same draws on the same stream.
    Natively (csrc/hostplan.hip pfrl_pyrandom_permutation: CPython's pool algorithm on the module's
    own MT19937 state, ~0.3 ms at n = 65 536 against 10 - 20 ms in the interpreter) when ``random``
    is the stock module-level generator, through the interpreter otherwise."""
    inst = getattr(random, "_inst", None)
    if (n >= 1024 and type(inst) is random.Random
            and getattr(random.sample, "__self__", None) is inst
            and getattr(random.getstate, "__self__", None) is inst):
        try:
            from pfrl_amd import _native

            lib = _native.lib()
            version, words, gauss = random.getstate()
            if version == 3 and len(words) == 625:
                state = np.array(words, dtype=np.uint32)
                out = np.empty(n, dtype=np.int64)
                _native.check(lib.pfrl_pyrandom_permutation(state.ctypes.data, int(n),
                                                            out.ctypes.data), "pyrandom_permutation")
                random.setstate((version, tuple(int(w) for w in state), gauss))
                return out
        except (RuntimeError, OSError, AttributeError):
            pass
    return np.asarray(random.sample(range(n), k=n), dtype=np.int64)


def _iter_minibatch_positions(n, minibatch_size, num_epochs):
    """:func:`_yield_minibatch_positions` as int64 arrays: the same ``random.sample`` draws in the
    same order (nothing else consumes Python's ``random`` during an update), the permutations
    drawn natively and kept as arrays -- converting a 16 384-element Python list per minibatch
    was 0.7 ms of host time in front of every update, drawing a permutation 10 - 20 ms in front
    of every epoch (profiles/r04_ppo_trace_summary.txt)."""
    buf = np.zeros(0, dtype=np.int64)
    done = 0
    while done < n * num_epochs:
        while len(buf) < minibatch_size:
            buf = np.concatenate([_random_permutation(n), buf])
        yield buf[len(buf) - minibatch_size:]
        done += minibatch_size
        buf = buf[:len(buf) - minibatch_size]


def _all_minibatch_positions(n, minibatch_size, num_epochs):
    return list(_iter_minibatch_positions(n, minibatch_size, num_epochs))


_HEAD_LOSS = os.environ.get("PFRL_PPO_HEAD_LOSS", "1") != "0"

class _Rollout:
    """T x N on-device rollout (env index minor)."""

    def __init__(self, device, n_envs, k, t_cap, act_shape, act_dtype):
        self.device = device
        self.N, self.k, self.cap = n_envs, k, t_cap
        self.T = 0
        self.h_state = np.zeros((t_cap, n_envs, k), dtype=np.int32)
        self.h_next = np.zeros((t_cap, n_envs, k), dtype=np.int32)
        self.h_reward = np.zeros((t_cap, n_envs), dtype=np.float64)
        self.h_nonterm = np.zeros((t_cap, n_envs), dtype=np.uint8)
        self.h_cut = np.zeros((t_cap, n_envs), dtype=np.uint8)
        self.h_action = np.zeros((t_cap, n_envs) + tuple(act_shape), dtype=act_dtype)
        self.d_action = None      # [t_cap, N, ...] on the device when the actions never left it
        self.closed = []          # (env, t_start, t_end) in completion order
        self.open_start = np.zeros(n_envs, dtype=np.int64)
        self.min_seq = None       # oldest frame-ring sequence number any stored ref points at

    def note_frames(self, min_seq):
        m = int(np.min(min_seq))
        self.min_seq = m if self.min_seq is None else min(self.min_seq, m)

    def add_step(self, s_refs, n_refs, action, reward, done, reset):
        t = self.T
        assert t < self.cap, "rollout longer than allocated"
        self.h_state[t] = s_refs
        self.h_next[t] = n_refs
        if isinstance(action, torch.Tensor):
            # device-resident actions (device env: nothing on the host ever looks at them)
            if self.d_action is None:
                self.d_action = torch.zeros((self.cap,) + tuple(action.shape), dtype=action.dtype,
                                            device=action.device)
                if t > 0:
                    self.d_action[:t].copy_(torch.from_numpy(self.h_action[:t]))
            if action.data_ptr() != self.d_action[t].data_ptr():    # (else: written there by the act graph)
                self.d_action[t].copy_(action)
        else:
            self.h_action[t] = action
            if self.d_action is not None:
                self.d_action[t].copy_(torch.from_numpy(np.asarray(action)))
        self.h_reward[t] = reward
        self.h_nonterm[t] = ~done
        end = done | reset
        self.h_cut[t] = end
        for e in np.flatnonzero(end):   # reference :786-789, env order
            self.closed.append((int(e), int(self.open_start[e]), t))
            self.open_start[e] = t + 1
        self.T = t + 1

    def size(self):
        return self.T * self.N

    def dataset_order(self):
        """Flat (t * N + e) index of every dataset position, in the reference's
        order: completed episodes first (completion order), then the unfinished
        fragments in env order (reference :450-458)."""
        parts = []
        T, N = self.T, self.N
        for e, a, b in self.closed:
            parts.append(np.arange(a, b + 1, dtype=np.int64) * N + e)
        for e in range(N):
            a = int(self.open_start[e])
            if a < T:
                parts.append(np.arange(a, T, dtype=np.int64) * N + e)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)

    def fragments(self):
        """(env, t_start, t_end inclusive) of every fragment in the reference's ``memory`` order:
        finished ones as they completed, then the open ones in env order (reference :448-456,
        :786-789)."""
        out = list(self.closed)
        for e in range(self.N):
            a = int(self.open_start[e])
            if a < self.T:
                out.append((e, a, self.T - 1))
        return out

    def reset(self):
        self.T = 0
        self.closed = []
        self.open_start[:] = 0
        self.min_seq = None


def model_tail(model, h):
    """The last child of a Sequential applied to what its other children produced."""
    return list(model._modules.values())[-1](h)


class _ActGraph:
    """The device side of ``batch_act`` during a rollout -- observation gather, network, sampling,
    entropy -- as ONE captured HIP graph per batch shape (reference ppo.py:759-778).

    A rollout step launches ~45 small kernels through the dispatcher (torch.distributions builds
    a Categorical, normalises logits, samples, takes the entropy: ~0.74 ms of host time per step
    for ~0.25 ms of device time, profiles/r04_ppo_trace_summary.txt: 100 ms of launch gaps per
    330 ms rollout).  Replayed, the step costs the host one small copy and one hipGraphLaunch.
    The graph reads the frame ring, the parameters and the normaliser statistics in place, so
    optimizer steps and new frames need no re-capture; sampling draws from the default generator
    through PyTorch's graph-safe Philox offsets (each replay advances the stream).

    Seed streams: with the example network the action is drawn by inverse CDF from ONE
    ``torch.rand`` per env (``pfrl_ppo_act_head``), not by ``Categorical.sample`` /
    ``torch.multinomial``, so for a given torch seed the action stream on a device env differs
    from the eager path's (same distribution: tests/test_hip_kernels.py::
    test_ppo_act_head_matches_torch_categorical; ``PFRL_PPO_ACT_GRAPH=0`` or
    ``PFRL_PPO_ACT_HEAD=0`` restore the eager draws).  The reference's trajectory parity fixtures
    (tests/golden/agent_trace_ppo.npz) are recorded with host envs and take the eager path."""

    def __init__(self, agent):
        self.agent = agent
        self.entries = {}
        self.pool = None

    def applicable(self):
        ag = self.agent
        return (os.environ.get("PFRL_PPO_ACT_GRAPH", "1") != "0" and ag.device.type == "cuda"
                and type(ag)._sample_action is PPO._sample_action
                and "_sample_action" not in ag.__dict__)

    def _split(self):
        """(body, policy layer, value layer) when the model is ``Sequential(..., Branched(Sequential(
        Linear(K, A), SoftmaxCategoricalHead()), Linear(K, 1)))`` -- the example network
        (examples/atari/train_ppo_ale.py:247-264): its two narrow heads, the sampling and the
        entropy then run as ONE launch (pfrl_ppo_act_head).  ``body`` is a view of the model without
        its last child (same class, same children and parameters, so a fused trunk stays fused)."""
        model = self.agent.model
        hit = self.__dict__.get("_split_cache")
        if hit is not None and hit[0] is model:
            return hit[1]
        import collections

        from pfrl_amd.nn import Branched
        from pfrl_amd.policies import SoftmaxCategoricalHead

        out = None
        nn = torch.nn
        if (os.environ.get("PFRL_PPO_ACT_HEAD", "1") != "0" and isinstance(model, nn.Sequential)
                and len(model) >= 2 and type(model[len(model) - 1]) is Branched):
            kids = list(model[len(model) - 1].child_modules)
            if len(kids) == 2 and type(kids[0]) is nn.Sequential and len(kids[0]) == 2:
                pol, head, val = kids[0][0], kids[0][1], kids[1]
                if (isinstance(pol, nn.Linear) and type(head) is SoftmaxCategoricalHead
                        and isinstance(val, nn.Linear) and val.out_features == 1
                        and pol.in_features == val.in_features and 1 <= pol.out_features <= 31
                        and pol.bias is not None and val.bias is not None
                        and pol.weight.dtype == torch.float32):
                    body = object.__new__(type(model))
                    body.__dict__ = dict(model.__dict__)
                    body._modules = collections.OrderedDict(list(model._modules.items())[:-1])
                    out = (body, pol, val)
        self._split_cache = (model, out)
        return out

    def _body(self, refs, into=None):
        ag = self.agent
        b_state = ag._features(refs)
        split = self._split()
        with torch.no_grad(), evaluating(ag.model):
            if split is not None:
                body, pol, val = split
                h = body(b_state)
                if h.dim() == 2 and h.dtype == torch.float32 and h.is_contiguous():
                    u = torch.rand(h.shape[0], dtype=torch.float32, device=h.device)
                    if into is not None:
                        # (action / entropy / value land in the rollout's own columns: no stack,
                        # no clones, no copy into the action column afterwards)
                        ops.ppo_act_head(h, pol.weight, pol.bias, val.weight, val.bias, u, into=into)
                        return None, None
                    action, entropy, value = ops.ppo_act_head(h, pol.weight, pol.bias, val.weight,
                                                              val.bias, u)
                    return action, torch.stack([entropy, value])
                distrib, value = model_tail(ag.model, h)
            else:
                distrib, value = ag.model(b_state)
            assert into is None
            action = distrib.sample()
            stats = torch.stack([distrib.entropy().reshape(-1).float(),
                                 value.reshape(-1).float()])
        return action, stats

    def in_place_ok(self, n_env):
        """The rollout-column form of a step: the fused head (``_split``), int64 actions [N]."""
        return (os.environ.get("PFRL_PPO_ACT_IN_PLACE", "1") != "0" and self._split() is not None
                and self.agent.obs_normalizer is None)

    def _capture(self, refs_dev, into=None):
        from pfrl_amd.agents.graphed_update import _capturing, _no_distribution_validation

        dev = self.agent.device
        if into is None:
            refs = refs_dev.clone()
            block = None
        else:
            # ONE input block: the step's frame slots + the two row indices, shipped straight from
            # the pinned staging slot (no device-side copy in front of the replay)
            nb = refs_dev.numel() * 4
            block = torch.zeros(((nb + 15) & ~15) + 16, dtype=torch.uint8, device=dev)
            refs = block[:nb].view(torch.int32).view(refs_dev.shape)
            refs.copy_(refs_dev)
            rows = block[(nb + 15) & ~15:][:8].view(torch.int32)
            # (warm-up and capture write somewhere nobody reads: the LAST row of the action column
            # -- the rollout never gets that far -- and the ring's reserved last slot; a capture in
            # the middle of a rollout must not touch rows that hold its data)
            rows.copy_(torch.tensor([into[0].shape[0] - 1, into[1].shape[0] - 1], dtype=torch.int32))
            into = (into[0], into[1], rows)
        rng = torch.cuda.get_rng_state(dev)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), _no_distribution_validation():
            for _ in range(2):
                self._body(refs, into)
        cur.wait_stream(side)
        torch.cuda.set_rng_state(rng, dev)      # the warm-up draws are not part of the run
        g = torch.cuda.CUDAGraph()
        with ops.profile_paused(), _capturing(g, self.pool), _no_distribution_validation():
            action, stats = self._body(refs, into)
        if self.pool is None:
            self.pool = g.pool()
        return g, refs, action, stats, block

    def run(self, refs_dev):
        """(actions [N], stats [2, N] = entropy, value): tensors OWNED BY THE GRAPH, overwritten by
        the next replay -- callers copy what they keep."""
        # (the graph bakes in the module tree it walked: a replaced child captures anew)
        key = (tuple(refs_dev.shape), tuple(id(m) for m in self.agent.model.modules()),
               self.agent.frames.emit_channels_last if self.agent.frames is not None else None)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = self._capture(refs_dev)
        g, refs, action, stats, _ = e
        refs.copy_(refs_dev)
        g.replay()
        return action, stats

    def run_in_place(self, refs_host, col, ring, row, slot, stage):
        """One rollout step whose outputs land in the rollout's columns: ``col`` [T, N] i64 gets
        the actions in row ``row``, ``ring`` [R, 2, N] f32 the (entropy, value) block in slot
        ``slot``.  The step costs ONE staging transfer (frame slots + the two indices, into the
        graph's own input block) and ONE replay.  Returns (col[row], ring[slot])."""
        if self.agent.frames is not None:
            from pfrl_amd.nn.atari_cnn import wants_channels_last

            # (the layout switch _gather() would flip inside the capture: settle it first, so that
            # the key of the second step is the key of the first)
            self.agent.frames.emit_channels_last = wants_channels_last(self.agent.model)
        key = (tuple(refs_host.shape), tuple(id(m) for m in self.agent.model.modules()),
               self.agent.frames.emit_channels_last if self.agent.frames is not None else None,
               col.data_ptr(), ring.data_ptr())
        e = self.entries.get(key)
        if e is None:
            (refs_dev,) = stage.upload([refs_host])
            e = self.entries[key] = self._capture(refs_dev, into=(col, ring))
        g, _, _, _, block = e
        stage.upload_to(block, [refs_host, np.array([row, slot], dtype=np.int32)])
        g.replay()
        return col[row], ring[slot]


class PPO(agent.AttributeSavingMixin, agent.BatchAgent):
    """Proximal Policy Optimization (arguments as in the reference)."""

    saved_attributes = ("model", "optimizer", "obs_normalizer")

    def __init__(self, model, optimizer, obs_normalizer=None, gpu=None, gamma=0.99, lambd=0.95,
                 phi=lambda x: x, value_func_coef=1.0, entropy_coef=0.01, update_interval=2048,
                 minibatch_size=64, epochs=10, clip_eps=0.2, clip_eps_vf=None,
                 standardize_advantages=True, batch_states=batch_states, recurrent=False,
                 max_recurrent_sequence_len=None, act_deterministically=False, max_grad_norm=None,
                 value_stats_window=1000, entropy_stats_window=1000, value_loss_stats_window=100,
                 policy_loss_stats_window=100, value_pass_chunk=16384,
                 reuse_next_values=False):
        self.model = model
        self.optimizer = optimizer
        self.obs_normalizer = obs_normalizer
        on_gpu = gpu is not None and gpu >= 0
        if on_gpu:
            assert torch.cuda.is_available()
            self.device = torch.device("cuda:{}".format(gpu))
            self.model.to(self.device)
            if self.obs_normalizer is not None:
                self.obs_normalizer.to(self.device)
            from pfrl_amd import _native

            _native.lib()          # no CPU fallback on this path: a missing library raises
        else:
            # the plumbing path of the reference: lists of transition dicts, stock torch ops
            self.device = torch.device("cpu")
        self.gamma = gamma
        self.lambd = lambd
        self.phi = phi
        self.value_func_coef = value_func_coef
        self.entropy_coef = entropy_coef
        self.update_interval = update_interval
        self.minibatch_size = minibatch_size
        self.epochs = epochs
        self.clip_eps = clip_eps
        self.clip_eps_vf = clip_eps_vf
        self.standardize_advantages = standardize_advantages
        self.batch_states = batch_states
        self.recurrent = bool(recurrent)
        self.max_recurrent_sequence_len = max_recurrent_sequence_len
        self.act_deterministically = act_deterministically
        self.max_grad_norm = max_grad_norm
        self.value_pass_chunk = value_pass_chunk
        # False (default): V over states AND next_states, as the reference (ppo.py:119-133).  On
        # the device path rows of the second pass that ARE rows of the first -- the next
        # observation of (t, env) is the observation of (t + 1, env) unless an episode ended --
        # are not evaluated twice where that is provably the same bits (_next_value_plan); the
        # result is the full second pass bit for bit (tests/test_bench_path_parity.py).
        # True (opt-in): the same shortcut WITHOUT the guarantee -- the remaining rows run as a
        # small batch of their own (other tile programs: values equal to f32 rounding only).
        self.reuse_next_values = bool(reuse_next_values)
        self.next_value_pass = None    # what the last rollout's second pass did (for bench.py)
        self.logger = getLogger(__name__)

        self.rollout = None
        self._act_graph = None
        self.device_actions = os.environ.get("PFRL_DEVICE_STEP", "1") != "0"
        self._last_action_dev = None
        self.ingest = None         # DeviceReplayStore used for host-observation ingestion
        self.frames = None
        self.batch_last_state = None
        self.batch_last_action = None
        self._last_refs = None

        self.value_record = _DeviceRecord(value_stats_window)
        self.entropy_record = _DeviceRecord(entropy_stats_window)
        self.value_loss_record = _DeviceRecord(value_loss_stats_window)
        self.policy_loss_record = _DeviceRecord(policy_loss_stats_window)
        self.explained_variance = np.nan
        self.n_updates = 0
        self._reward_mode = None
        from pfrl_amd.distributed import GradientAllReducer

        self.grad_reducer = GradientAllReducer(self.model)
        self._host = None
        self._rec = None
        if on_gpu and not (recurrent and os.environ.get("PFRL_PPO_RECURRENT_HOST") == "1"):
            from pfrl_amd.staging import StagingRing

            self._stage = StagingRing(self.device,
                                      slot_bytes=max(1 << 22, 96 * int(update_interval)),
                                      n_slots=8)
            if recurrent:
                # the same HBM rollout columns + two columns of recurrent states; fragments and
                # sequences are arrays of positions (agents/_ppo_recurrent_device.py; reference
                # ppo.py:56-107,534-632)
                from pfrl_amd.agents._ppo_recurrent_device import RecurrentDeviceRollouts

                self._rec = RecurrentDeviceRollouts(self)
        else:
            # gpu=None (and PFRL_PPO_RECURRENT_HOST=1, the round-5 arrangement for recurrent
            # models: network on the device, rollout bookkeeping on the host): the reference's
            # fragments of transition dicts
            from pfrl_amd.agents.ppo_host import HostRollouts

            self._host = HostRollouts(self)

    @property
    def memory(self):
        """Finished fragments of the current rollout as lists of transition dicts -- host path
        only (reference attribute; the device path keeps columns in HBM, see ``rollout``)."""
        if self._host is None:
            raise AttributeError("PPO.memory exists on the host path only (gpu=None)")
        return self._host.memory

    # -- observations ------------------------------------------------------------
    def _refs_of(self, batch_obs):
        """Frame slots [N, k] of a batch of observations (device or host)."""
        if isinstance(batch_obs, DeviceObsBatch):
            if self.frames is None:
                self.frames = batch_obs.store
            return batch_obs.refs, batch_obs
        if self.ingest is None:
            from pfrl_amd.replay_buffers.device_replay import DeviceReplayStore

            self.ingest = DeviceReplayStore(
                self.device, capacity=self.update_interval + 4 * len(batch_obs) + 64, num_steps=1)
            self.ingest.set_phi(self.phi)
        pairs = [self.ingest.ingest(o) for o in batch_obs]
        self.ingest.flush()
        self.frames = self.ingest.frames
        refs = np.stack([p[0] for p in pairs]).astype(np.int32)
        batch = DeviceObsBatch(self.frames, refs, np.array([p[1] for p in pairs]))
        return refs, batch

    def _divisor(self):
        if self.ingest is not None:
            return self.ingest.divisor_for(self.phi)
        from pfrl_amd.utils.batch_states import _divisor_for

        d = _divisor_for(self.phi, lambda: self._sample_obs.to_numpy())
        if d is None:
            raise TypeError("pfrl_amd.PPO: phi must be a cast/scale feature extractor")
        return d

    def _gather(self, refs_dev):
        from pfrl_amd.nn.atari_cnn import wants_channels_last

        self.frames.emit_channels_last = wants_channels_last(self.model)
        x = self.frames.gather(refs_dev, self._divisor())
        if x.dim() == 4 and not x.is_contiguous():
            return x    # channels_last [M, 4, H, W]: already the network's input
        fs = self.frames.frame_shape
        if refs_dev.shape[1] == 1:
            return x.view((x.shape[0],) + fs)
        if len(fs) >= 2 and fs[0] == 1:
            return x.view((x.shape[0], refs_dev.shape[1]) + fs[1:])
        return x

    def _u8_pixels(self, refs_dev):
        """The observations as u8 NHWC4 pixels (ops.U8Pixels) where the model's first stage is the
        MFMA trunk and its first convolution can evaluate ``phi(x) = float32(x) / d`` in its own
        operand loader (nn/mfma_trunk.py ``u8_first_layer_shape_ok``), else None.  The batch is
        then gathered with 2 bytes moved per frame byte instead of 5, and the layer's forward and
        weight-gradient launches read a quarter of the bytes -- with bit-identical activations
        and gradients (tests/test_mfma_trunk.py).  ``PFRL_U8_CONV1=0`` keeps the fp32 batch."""
        from pfrl_amd.nn import mfma_trunk
        from pfrl_amd.nn.atari_cnn import wants_channels_last

        fr = self.frames
        if not (self.obs_normalizer is None and isinstance(self.model, mfma_trunk._TrunkSequential)
                and self.model._trunk_run[0] == 0 and refs_dev.dim() == 2 and refs_dev.shape[1] == 4
                and refs_dev.is_contiguous() and fr.frames.dtype == torch.uint8
                and wants_channels_last(self.model) and ops.channels_last_supported(fr.frames, 4)):
            return None
        first = list(self.model._modules.values())[self.model._trunk_run[2][0]]
        hw = fr.frames.shape[-2:]
        if not mfma_trunk.u8_first_layer_shape_ok(first, refs_dev.shape[0], hw[0], hw[1],
                                                  self._divisor()):
            return None
        return ops.batch_states_raw_nhwc4(fr.frames, refs_dev, self._divisor())

    def _features(self, refs_dev):
        """Network input for a batch of observation refs: the gathered fp32 batch,
        normalised with the CURRENT statistics of ``obs_normalizer`` if there is one
        (reference ppo.py:75-77,124-126,486-487,689-690: always ``update=False``; the
        statistics only learn in :meth:`_update`, once per rollout)."""
        px = self._u8_pixels(refs_dev)
        if px is not None:
            return px       # (the trunk's first convolution applies phi itself)
        x = self._gather(refs_dev)
        if self.obs_normalizer is not None:
            x = self.obs_normalizer(x, update=False)
        return x

    # -- acting --------------------------------------------------------------------
    def _sample_action(self, action_distrib):
        return action_distrib.sample()

    def _batch_act_train(self, batch_obs):
        assert self.training
        refs, dev_batch = self._refs_of(batch_obs)
        self._sample_obs = dev_batch[0]
        if self._act_graph is None:
            self._act_graph = _ActGraph(self)
        in_place = (self._rec is None and isinstance(batch_obs, DeviceObsBatch) and self.device_actions
                    and self._act_graph.applicable() and self._act_graph.in_place_ok(len(batch_obs)))
        refs_dev = None if in_place else self._stage.upload([refs])[0]
        if self._rec is not None:
            action_dev = self._rec.act_train(refs_dev)
        elif (isinstance(batch_obs, DeviceObsBatch) and self.device_actions
                and self._act_graph.applicable() and self._act_graph.in_place_ok(len(batch_obs))):
            # device env, example network: ONE transfer + ONE replay per step; the actions land in
            # the rollout's action column, entropy / value in a ring the statistics windows read
            ro = self._ensure_rollout(len(batch_obs), refs.shape[1], (), np.dtype(np.int64))
            if ro.d_action is None:
                ro.d_action = torch.zeros((ro.cap, ro.N), dtype=torch.int64, device=self.device)
            ring = self._stats_ring(len(batch_obs))
            slot = self._stats_at % (ring.shape[0] - 1)      # (the last slot: the capture's scratch)
            self._stats_at += 1
            action_dev, stats = self._act_graph.run_in_place(refs, ro.d_action, ring, ro.T, slot,
                                                             self._stage)
            self.entropy_record.extend(stats[0])
            self.value_record.extend(stats[1])
        elif (isinstance(batch_obs, DeviceObsBatch) and self.device_actions
                and self._act_graph.applicable()):
            # device env: nothing of this step is looked at on the host -- one graph replay
            action_dev, stats = self._act_graph.run(refs_dev)
            action_dev = action_dev.clone()
            stats = stats.clone()
            self.entropy_record.extend(stats[0])
            self.value_record.extend(stats[1])
        else:
            b_state = self._features(refs_dev)
            with torch.no_grad(), evaluating(self.model):
                action_distrib, batch_value = self.model(b_state)
                action_dev = self._sample_action(action_distrib)
                self.entropy_record.extend(action_distrib.entropy())
                self.value_record.extend(batch_value)
        self._last_refs = refs.copy()
        self._last_min_seq = int(np.min(dev_batch.min_seq))
        self.batch_last_state = list(range(len(batch_obs)))
        if isinstance(batch_obs, DeviceObsBatch) and self.device_actions:
            # a device env: the sampled actions stay in HBM (one D2H only if the env or the
            # driver looks at them), the rollout takes its action column from this tensor and
            # the host never waits for the acting forward pass
            from pfrl_amd.device_store import DeviceActions

            self._last_action_dev = action_dev
            self.batch_last_action = DeviceActions(action_dev)
            return self.batch_last_action
        self._last_action_dev = None
        batch_action = action_dev.cpu().numpy()
        self.batch_last_action = list(batch_action)
        return batch_action

    def _batch_act_eval(self, batch_obs):
        assert not self.training
        if isinstance(batch_obs, DeviceObsBatch) or (
                len(batch_obs) > 0 and isinstance(batch_obs[0], DeviceObs)):
            refs, dev_batch = self._refs_of(batch_obs)
            self._sample_obs = dev_batch[0]
            (refs_dev,) = self._stage.upload([refs])
            b_state = self._features(refs_dev)
        else:
            # Host observations of an evaluation episode are uploaded directly (the
            # reference's batch_states, pfrl/agents/ppo.py:689-690): they must not take
            # slots of the frame ring the pending rollout's transitions point into.
            b_state = self.batch_states(batch_obs, self.device, self.phi)
            if self.obs_normalizer is not None:
                b_state = self.obs_normalizer(b_state, update=False)
        with torch.no_grad(), evaluating(self.model):
            if self._rec is not None:
                action_distrib = self._rec.act_eval(b_state)
            else:
                action_distrib, _ = self.model(b_state)
            if self.act_deterministically:
                action = mode_of_distribution(action_distrib).cpu().numpy()
            else:
                action = action_distrib.sample().cpu().numpy()
        return action

    _stats_at = 0
    _stats_ring_buf = None

    def _stats_ring(self, n_env):
        """[R, 2, N] f32: (entropy, value) of the last R acting steps, R large enough that a slot
        is reused only after both statistics windows have dropped it."""
        window = max(self.value_record.maxlen, self.entropy_record.maxlen)
        R = -(-window // n_env) + 4
        ring = self._stats_ring_buf
        if ring is None or tuple(ring.shape) != (R, 2, n_env):
            ring = self._stats_ring_buf = torch.zeros((R, 2, n_env), dtype=torch.float32,
                                                      device=self.device)
        return ring

    def _ensure_rollout(self, n_env, k, act_shape, act_dtype):
        if self.rollout is None:
            t_cap = -(-self.update_interval // n_env) + 2
            self.rollout = _Rollout(self.device, n_env, k, t_cap, act_shape, act_dtype)
        return self.rollout

    def batch_act(self, batch_obs):
        if self._host is not None:
            act = self._host.batch_act_train if self.training else self._host.batch_act_eval
            return act(batch_obs)
        if self.training:
            return self._batch_act_train(batch_obs)
        return self._batch_act_eval(batch_obs)

    # -- observing -------------------------------------------------------------------
    def _batch_observe_train(self, batch_obs, batch_reward, batch_done, batch_reset):
        assert self.training
        n_env = len(batch_obs)
        next_refs, next_batch = self._refs_of(batch_obs)
        actions = self.__dict__.get("_last_action_dev")
        if actions is None:
            actions = np.asarray(self.batch_last_action)
        self._ensure_rollout(n_env, next_refs.shape[1], actions.shape[1:],
                             actions.dtype if isinstance(actions, np.ndarray) else
                             np.dtype(str(actions.dtype).replace("torch.", "")))
        if self._reward_mode is None:
            r0 = batch_reward[0]
            # NEP 50: np.float64 rewards promote the GAE arithmetic to f64,
            # Python floats / np.float32 keep it in f32
            self._reward_mode = 1 if isinstance(r0, np.float64) else 0
        done = np.asarray(batch_done, dtype=bool)
        reset = np.asarray(batch_reset, dtype=bool)
        self.rollout.add_step(self._last_refs, next_refs, actions,
                              np.asarray(batch_reward, dtype=np.float64), done, reset)
        if self._rec is not None:
            self._rec.observe_train(self.rollout.T - 1, done, reset)
        self.rollout.note_frames(min(self._last_min_seq, int(np.min(next_batch.min_seq))))
        self.batch_last_state = [None] * n_env
        self.batch_last_action = [None] * n_env
        self._last_action_dev = None
        self._update_if_dataset_is_ready()

    def batch_observe(self, batch_obs, batch_reward, batch_done, batch_reset):
        if self._host is not None:
            observe = (self._host.batch_observe_train if self.training
                       else self._host.batch_observe_eval)
            return observe(batch_obs, batch_reward, batch_done, batch_reset)
        if self.training:
            self._batch_observe_train(batch_obs, batch_reward, batch_done, batch_reset)
        elif self._rec is not None:
            self._rec.observe_eval(batch_done, batch_reset)

    # -- learning --------------------------------------------------------------------
    def _update_if_dataset_is_ready(self):
        if self.rollout.size() >= self.update_interval:
            self._update()
            self.rollout.reset()

    def _value_pass(self, refs_dev, actions_dev):
        """log pi(a|s) and V(s) for every rollout position (reference :110-142),
        chunked so that the fp32 observation batch stays a few hundred MB."""
        M = refs_dev.shape[0]
        log_probs = torch.empty(M, dtype=torch.float32, device=self.device)
        values = torch.empty(M, dtype=torch.float32, device=self.device)
        split = None
        if self._act_graph is not None and self._act_graph.applicable() and (
                actions_dev is None or actions_dev.dtype == torch.int64):
            split = self._act_graph._split()
        with torch.no_grad(), evaluating(self.model):
            for lo in range(0, M, self.value_pass_chunk):
                hi = min(M, lo + self.value_pass_chunk)
                x = self._features(refs_dev[lo:hi])
                if split is not None:
                    # heads + log pi(a | s) + V(s) in one launch behind the trunk (the acting path's
                    # kernel without the draw), written straight into the columns
                    body, pol, val = split
                    h = body(x)
                    if h.dim() == 2 and h.dtype == torch.float32 and h.is_contiguous():
                        ops.ppo_value_head(h, pol.weight, pol.bias, val.weight, val.bias,
                                           actions_dev[lo:hi] if actions_dev is not None else None,
                                           log_probs[lo:hi], values[lo:hi])
                        continue
                    distribs, vs = model_tail(self.model, h)
                else:
                    distribs, vs = self.model(x)
                values[lo:hi] = vs.reshape(-1)
                if actions_dev is not None:
                    log_probs[lo:hi] = distribs.log_prob(actions_dev[lo:hi])
        return log_probs, values

    def _next_value_plan(self, M):
        """Rows per chunk of the value pass if evaluating a FEW rows of it on their own can be made
        to give bit for bit what the whole pass gives, else 0.  Holds when everything between the
        frame ring and V(s) is this library's row-independent kernels (u8 / fp32 gather, MFMA
        trunk, the fused value head: no launch mixes rows, and ``mfma_trunk.plan_batch`` pins the
        tile programs, split-K and layout route to the ones a full chunk takes) and every chunk
        of the pass has the same size.  Library GEMMs / convolutions (another model, an
        ``obs_normalizer``) pick kernels by batch size: no guarantee, the full pass runs."""
        from pfrl_amd.nn import mfma_trunk

        if os.environ.get("PFRL_PPO_DEDUP_NEXT", "1") == "0" or self.obs_normalizer is not None:
            return 0
        chunk = int(self.value_pass_chunk)
        if not (M <= chunk or M % chunk == 0):
            return 0
        if self._act_graph is None:
            self._act_graph = _ActGraph(self)
        split = self._act_graph._split() if self._act_graph.applicable() else None
        if split is None:
            return 0
        body = split[0]
        run = getattr(body, "_trunk_run", None)
        if not (isinstance(body, mfma_trunk._TrunkSequential) and run is not None and run[0] == 0
                and run[1] == len(body._modules)):
            return 0
        return min(M, chunk)

    def _next_values(self, ro, T, N, v_pred, n_refs):
        """V(next_state) for every rollout position (reference ppo.py:119-133)."""
        M = T * N
        if self.reuse_next_values:
            self.next_value_pass = {"mode": "reuse_next_values (opt-in, f32-rounding equal)", "of": M}
            return self._next_values_from_states(ro, T, N, v_pred, n_refs)
        rows = self._next_value_plan(M)
        if rows:
            from pfrl_amd.nn import mfma_trunk

            self.next_value_pass = {"mode": "rows shared with the state pass evaluated once "
                                            "(bit-identical to the full pass)", "of": M}
            with mfma_trunk.plan_batch(rows):
                return self._next_values_from_states(ro, T, N, v_pred, n_refs)
        self.next_value_pass = {"mode": "full pass", "of": M, "evaluated": M}
        _, next_v = self._value_pass(n_refs, None)
        return next_v

    def _next_values_from_states(self, ro, T, N, v_pred, n_refs):
        """V(next_state) without a second pass over the whole rollout: wherever the
        next observation of (t, env) IS the observation of (t+1, env) -- the same
        frame slots, i.e. no episode end / reset in between -- its value is
        v_pred[t+1, env], computed by the same network in the same mode.  Only the
        other rows (episode ends, and the last step of the rollout) are evaluated."""
        same = (ro.h_next[:T - 1] == ro.h_state[1:T]).all(axis=-1) if T > 1 else \
            np.zeros((0, N), dtype=bool)
        need = np.ones((T, N), dtype=bool)
        need[:T - 1] = ~same
        need_idx = np.flatnonzero(need.reshape(-1))
        # pad to a multiple of N with repeats: a handful of distinct batch shapes
        # instead of a new one (and a new MIOpen solver lookup) every rollout
        pad = (-len(need_idx)) % N
        if pad:
            need_idx = np.concatenate([need_idx, np.repeat(need_idx[-1:], pad)])
        if self.next_value_pass is not None:
            self.next_value_pass["evaluated"] = int(len(need_idx))
        (idx_dev,) = self._stage.upload([need_idx.astype(np.int64)])
        idx_dev = idx_dev.clone()
        _, vals = self._value_pass(n_refs[idx_dev], None)
        next_v = torch.empty_like(v_pred)
        if T > 1:
            next_v[:(T - 1) * N] = v_pred[N:]
        next_v[idx_dev] = vals
        return next_v

    def _check_frames_alive(self, ro):
        if (ro.min_seq is not None and self.frames is not None
                and ro.min_seq < self.frames.oldest_live_seq()):
            # same liveness rule as the replay store's slots_for(): a rollout whose oldest
            # frame has been overwritten would train on other observations, silently
            raise RuntimeError(
                "PPO rollout refers to frame %d but the frame ring (%d slots) has wrapped past "
                "it (oldest live frame %d); give the frame store more slots than one rollout "
                "writes" % (ro.min_seq, self.frames.n_slots, self.frames.oldest_live_seq()))

    def _update(self):
        if self._rec is not None:
            return self._rec.update()
        ro = self.rollout
        T, N, k = ro.T, ro.N, ro.k
        dev = self.device
        order = ro.dataset_order()
        n = len(order)
        assert n == T * N
        self._check_frames_alive(ro)
        # ship the rollout columns (one transfer)
        on_dev = ro.d_action is not None
        up = self._stage.upload([
            ro.h_state[:T].reshape(T * N, k), ro.h_next[:T].reshape(T * N, k),
            (np.zeros(1, dtype=ro.h_action.dtype) if on_dev else
             ro.h_action[:T].reshape((T * N,) + ro.h_action.shape[2:])),
            ro.h_reward[:T].reshape(-1), ro.h_nonterm[:T].reshape(-1),
            self._cut_with_rollout_end(ro, T).reshape(-1), order])
        # staging views are recycled when the ring wraps (the minibatch loop below
        # uploads through the same ring): keep private device copies
        s_refs, n_refs, actions, reward, nonterm, cut, order_dev = [t.clone() for t in up]
        if on_dev:
            # the action column never left the device
            actions = ro.d_action[:T].reshape((T * N,) + tuple(ro.d_action.shape[2:])).clone()
        log_probs, v_pred = self._value_pass(s_refs, actions)
        minibatches = (order[pos] for pos in
                       _iter_minibatch_positions(n, self.minibatch_size, self.epochs))
        next_v = self._next_values(ro, T, N, v_pred, n_refs)
        adv, v_teacher = ops.gae_scan(reward.view(T, N), v_pred.view(T, N), next_v.view(T, N),
                                      nonterm.view(T, N), cut.view(T, N), self.gamma, self.lambd,
                                      self._reward_mode)
        adv = adv.view(-1)
        v_teacher = v_teacher.view(-1)
        if self.standardize_advantages:
            from pfrl_amd.distributed import global_mean_std

            mean_std = global_mean_std(ops.adv_stats(adv), n)
        else:
            mean_std = torch.zeros(2, dtype=torch.float32, device=dev)
        if self.obs_normalizer is not None:
            # the statistics learn from all states of the rollout at once, after the
            # value pass and before the epochs (reference ppo.py:460-471)
            with torch.no_grad():
                self.obs_normalizer.experience(self._gather(s_refs))
        actions_i64 = actions if actions.dtype == torch.int64 else None
        self._last_dataset = dict(order=order, adv=adv, v_teacher=v_teacher, v_pred=v_pred,
                                  log_prob=log_probs, mean_std=mean_std)

        captured = self._captured_update_ok(n, actions_i64)
        if captured:
            cols = self._static_columns(adv, mean_std, log_probs, v_pred, v_teacher, actions_i64,
                                        s_refs)
        for flat in minibatches:
            (idx,) = self._stage.upload([flat])
            if captured:
                # one graph replay per minibatch: gather, forward, loss, backward, clip, Adam --
                # the ~100 small launches of the heads and the loss no longer wait for the
                # dispatcher (1.8 ms -> 0.45 ms of a 12 ms update, profiles/r04_ppo_update_timeline.txt)
                cols["idx"].copy_(idx)
                out = self._update_graph.run({"idx": cols["idx"]}, baked=self._baked_hyperparameters())
                self.value_loss_record.extend(out["value_loss"].clone())
                self.policy_loss_record.extend(out["policy_loss"].clone())
                self.n_updates += 1
                continue
            if actions_i64 is not None:
                mb = ops.ppo_minibatch(idx, adv, mean_std, self.standardize_advantages, log_probs,
                                       v_pred, v_teacher, actions_i64, s_refs)
                mb_actions = mb["action"]
            else:
                dummy = torch.zeros(1, dtype=torch.int64, device=dev).expand(n).contiguous()
                mb = ops.ppo_minibatch(idx, adv, mean_std, self.standardize_advantages, log_probs,
                                       v_pred, v_teacher, dummy, s_refs)
                mb_actions = actions[idx]
            states = self._features(mb["refs"])
            distribs, vs_pred = self.model(states)
            self.model.zero_grad()
            loss = self._lossfun(
                distribs.entropy(), vs_pred, distribs.log_prob(mb_actions),
                vs_pred_old=mb["v_pred"][..., None], log_probs_old=mb["log_prob"],
                advs=mb["adv"], vs_teacher=mb["v_teacher"][..., None])
            loss.backward()
            self.grad_reducer.all_reduce()
            if self.max_grad_norm is not None:
                clip_grad_norm_device_(self.model.parameters(), self.max_grad_norm)
            self.optimizer.step()
            self.n_updates += 1
        # explained variance (reference :181-193), one small reduction
        with torch.no_grad():
            vart = torch.var(v_teacher, unbiased=False)
            ev = 1 - torch.var(v_teacher - v_pred, unbiased=False) / vart
            self.explained_variance = float("nan") if float(vart) == 0 else float(ev)

    # -- one minibatch update as one captured graph -------------------------------------------
    _update_graph = None
    _static_cols = None

    def _captured_update_ok(self, n, actions_i64):
        """Discrete actions, one process, a stock ``_lossfun``: the minibatch update is a fixed
        launch sequence on fixed-size tensors and can be replayed from a HIP graph."""
        from pfrl_amd import distributed

        return (os.environ.get("PFRL_PPO_UPDATE_GRAPH", "1") != "0" and self.device.type == "cuda"
                and actions_i64 is not None and self._dp_update_capturable()
                and n % self.minibatch_size == 0
                and type(self)._lossfun is PPO._lossfun and "_lossfun" not in self.__dict__)

    def _dp_update_capturable(self):
        """Data parallel (env-sharded rollouts): the gradient all-reduce of a
        minibatch sits between ``backward`` and ``clip`` + ``step``.  With the directly driven RCCL
        communicator it is stream-ordered and -- issued on the capturing stream itself,
        distributed.GradientAllReducer._exchange_stream -- a node of the captured update like any
        kernel, where a probe every rank takes part in says captured collectives replay
        (``captured_collectives_work``; ``PFRL_GRAPH_COLLECTIVE=0`` = bench.py's more conservative
        plans: the eager update with the eager collective)."""
        from pfrl_amd import distributed

        red = self.grad_reducer
        if not red.active():
            return True
        if os.environ.get("PFRL_GRAPH_COLLECTIVE", "auto") == "0":
            return False
        if red._comm is None and torch.distributed.get_backend() != "nccl":
            return False
        return distributed.captured_collectives_work(self.device)

    def _baked_hyperparameters(self):
        """What ``_minibatch_step`` / ``_lossfun`` read as Python numbers, i.e. what a captured
        update holds as kernel arguments: part of the graph key, so that a hook that moves one of
        them (the reference's ``LinearInterpolationHook`` on ``clip_eps``,
        examples/atari/train_ppo_ale.py:301-306) is not replayed away."""
        def num(v):
            return None if v is None else float(v)

        return (num(self.clip_eps), num(self.clip_eps_vf), num(self.entropy_coef),
                num(self.value_func_coef), num(self.max_grad_norm), bool(self.standardize_advantages),
                id(self.model), tuple(id(m) for m in self.model.children()))

    def _static_columns(self, adv, mean_std, log_probs, v_pred, v_teacher, actions_i64, s_refs):
        """The rollout's columns in buffers that keep their addresses from rollout to rollout
        (what the captured update reads), refreshed with one copy each per rollout."""
        src = dict(adv=adv, mean_std=mean_std, log_prob=log_probs, v_pred=v_pred,
                   v_teacher=v_teacher, action=actions_i64, s_refs=s_refs)
        cols = self._static_cols
        if cols is None or any(cols[k].shape != v.shape or cols[k].dtype != v.dtype
                               for k, v in src.items()):
            cols = self._static_cols = {k: torch.empty_like(v) for k, v in src.items()}
            cols["idx"] = torch.empty(self.minibatch_size, dtype=torch.int64, device=self.device)
            self._update_graph = None
        for k, v in src.items():
            cols[k].copy_(v)
        if self._update_graph is None:
            from pfrl_amd.agents.graphed_update import CapturedStep

            # (the learning rate in a device scalar: an lr schedule does not re-capture)
            self._update_graph = CapturedStep(self._minibatch_step, [self.model], [self.optimizer],
                                              self.device, lr_on_device=True)
        return cols

    def _fused_loss_split(self):
        """(body, policy layer, value layer) when the loss of a minibatch can run as the fused
        launch: the example network's two narrow heads behind one body (``_ActGraph._split``),
        the stock ``_lossfun``; ``PFRL_PPO_FUSED_LOSS=0`` keeps torch.distributions + autograd."""
        if (os.environ.get("PFRL_PPO_FUSED_LOSS", "1") == "0" or self.device.type != "cuda"
                or type(self)._lossfun is not PPO._lossfun or "_lossfun" in self.__dict__):
            return None
        if self._act_graph is None:
            self._act_graph = _ActGraph(self)
        return self._act_graph._split()

    def _minibatch_step(self, batch):
        """reference ppo.py:480-532 for one minibatch, on the static columns."""
        c = self._static_cols
        with ops.profile_paused():
            mb = ops.ppo_minibatch(batch["idx"], c["adv"], c["mean_std"], self.standardize_advantages,
                                   c["log_prob"], c["v_pred"], c["v_teacher"], c["action"],
                                   c["s_refs"])
            states = self._features(mb["refs"])
        fused = self._fused_loss_split()
        if fused is not None:
            # heads as plain layers on the trunk's output, then loss + its gradient with respect to
            # logits and values in ONE launch (pfrl_ppo_loss) instead of ~90 through
            # torch.distributions + autograd; backward starts at the logits / values
            body, pol, val = fused
            h = body(states)
            if _HEAD_LOSS and ops.ppo_head_loss_ok(h, pol.weight):
                # ... and the heads themselves, forward and backward, in the same launch: h is read
                # once and dh written once where the library route runs five narrow GEMMs and two
                # adds over the 16 384 rows (pfrl_ppo_head_loss; PFRL_PPO_HEAD_LOSS=0 keeps them)
                self.optimizer.zero_grad(set_to_none=True)
                out4, dh, (dwp, dbp, dwv, dbv) = ops.ppo_head_loss(
                    h, pol.weight, pol.bias, val.weight, val.bias, mb["action"], mb["adv"],
                    mb["log_prob"], mb["v_pred"], mb["v_teacher"], self.clip_eps, self.clip_eps_vf,
                    self.value_func_coef, self.entropy_coef)
                pol.weight.grad, pol.bias.grad, val.weight.grad, val.bias.grad = dwp, dbp, dwv, dbv
                h.backward(dh)
            else:
                logits, vs_pred = pol(h), val(h)
                self.optimizer.zero_grad(set_to_none=True)
                out4, dlogits, dvalue = ops.ppo_loss(
                    logits, vs_pred, mb["action"], mb["adv"], mb["log_prob"], mb["v_pred"],
                    mb["v_teacher"], self.clip_eps, self.clip_eps_vf, self.value_func_coef,
                    self.entropy_coef)
                torch.autograd.backward([logits, vs_pred], [dlogits, dvalue])
            loss = out4[0]
            records = {"value_loss": out4[2], "policy_loss": out4[1]}
        else:
            distribs, vs_pred = self.model(states)
            self.optimizer.zero_grad(set_to_none=True)
            records = {}
            loss = self._lossfun(
                distribs.entropy(), vs_pred, distribs.log_prob(mb["action"]),
                vs_pred_old=mb["v_pred"][..., None], log_probs_old=mb["log_prob"],
                advs=mb["adv"], vs_teacher=mb["v_teacher"][..., None], records=records)
            loss.backward()
        # (data parallel: one flat all-reduce per minibatch, a graph node when captured;
        # nothing without a process group)
        self.grad_reducer.all_reduce()
        if self.max_grad_norm is not None:
            clip_grad_norm_device_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return {"loss": loss.detach(), "value_loss": records["value_loss"].detach(),
                "policy_loss": records["policy_loss"].detach()}

    @staticmethod
    def _cut_with_rollout_end(ro, T):
        cut = ro.h_cut[:T].copy()
        cut[T - 1] = 1   # unfinished fragments end with the rollout (reference :450-458)
        return cut

    def _lossfun(self, entropy, vs_pred, log_probs, vs_pred_old, log_probs_old, advs, vs_teacher,
                 records=None):
        prob_ratio = torch.exp(log_probs - log_probs_old)
        loss_policy = -torch.mean(torch.min(
            prob_ratio * advs,
            torch.clamp(prob_ratio, 1 - self.clip_eps, 1 + self.clip_eps) * advs))
        if self.clip_eps_vf is None:
            loss_value_func = F.mse_loss(vs_pred, vs_teacher)
        else:
            clipped_vs_pred = _elementwise_clip(vs_pred, vs_pred_old - self.clip_eps_vf,
                                                vs_pred_old + self.clip_eps_vf)
            loss_value_func = torch.mean(torch.max(
                F.mse_loss(vs_pred, vs_teacher, reduction="none"),
                F.mse_loss(clipped_vs_pred, vs_teacher, reduction="none")))
        loss_entropy = -torch.mean(entropy)
        if records is None:
            self.value_loss_record.extend(loss_value_func)
            self.policy_loss_record.extend(loss_policy)
        else:
            # (a captured update: the tensors belong to the graph, the caller records copies)
            records["value_loss"], records["policy_loss"] = loss_value_func, loss_policy
        return (loss_policy + self.value_func_coef * loss_value_func
                + self.entropy_coef * loss_entropy)

    def get_statistics(self):
        return [
            ("average_value", _mean_or_nan(self.value_record.values())),
            ("average_entropy", _mean_or_nan(self.entropy_record.values())),
            ("average_value_loss", _mean_or_nan(self.value_loss_record.values())),
            ("average_policy_loss", _mean_or_nan(self.policy_loss_record.values())),
            ("n_updates", self.n_updates),
            ("explained_variance", self.explained_variance),
        ]
