"""Proportional prioritized replay (https://arxiv.org/abs/1511.05952, 3.3).

Mirrors ``pfrl.replay_buffers.prioritized``
(/root/reference/pfrl/replay_buffers/prioritized.py): ``PriorityWeightError``
(:9-66) and ``PrioritizedReplayBuffer`` (:69-126) with the same constructor
signature and defaults.  The sum / min trees, the B sequentially dependent
draws, the importance weights and the priority update all run on the device
(pfrl_amd/csrc/sumtree.hip); per ``sample`` the host contributes only the B
uniform draws taken from the global NumPy stream (so seeds mean the same thing
as in the reference) and per ``update_errors`` nothing at all when the TD
errors are already a device tensor.
"""

import numpy as np
import torch

from pfrl_amd.replay_buffer import DeviceExperienceBatch
from pfrl_amd.replay_buffers.replay_buffer import ReplayBuffer

_NORMALIZE_CODE = {False: 0, "batch": 1, "memory": 2}


class PriorityWeightError(object):
    """alpha / beta / eps arithmetic of proportional prioritisation."""

    def __init__(self, alpha, beta0, betasteps, eps, normalize_by_max, error_min, error_max):
        assert 0.0 <= alpha
        assert 0.0 <= beta0 <= 1.0
        self.alpha = alpha
        self.beta = beta0
        self.beta_add = 0 if betasteps is None else (1.0 - beta0) / betasteps
        self.eps = eps
        if normalize_by_max is True:
            normalize_by_max = "batch"
        assert normalize_by_max in [False, "batch", "memory"]
        self.normalize_by_max = normalize_by_max
        self.error_min = error_min
        self.error_max = error_max

    def _clip(self, error):
        if self.error_min is not None:
            error = max(self.error_min, error)
        if self.error_max is not None:
            error = min(self.error_max, error)
        return error

    def priority_from_errors(self, errors):
        """(clip(d) + eps) ** alpha for host scalars (reference :47-55); the
        scalar types (Python float / np.float32) are preserved exactly as NumPy
        would, because they end up inside the priority trees."""
        return [(self._clip(d) + self.eps) ** self.alpha for d in errors]

    def weights_from_probabilities(self, probabilities, min_probability):
        """Host version of the importance weights (reference :57-66)."""
        if self.normalize_by_max == "batch":
            min_probability = np.min(probabilities)
        if self.normalize_by_max:
            weights = [(p / min_probability) ** -self.beta for p in probabilities]
        else:
            weights = [(len(self.memory) * p) ** -self.beta for p in probabilities]
        self.beta = min(1.0, self.beta + self.beta_add)
        return weights


class _LazySeqs:
    """Entry sequence numbers of a prioritized sample, fetched from the device
    only if somebody asks for host views."""

    def __init__(self, x_dev, n):
        self._x = x_dev
        self._n = n
        self._host = None

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if self._host is None:
            torch.cuda.synchronize(self._x.device)   # may have been drawn on a replay stream
            self._host = self._x.cpu().numpy()
        return self._host[i]


class _DevicePrioritizedQueue:
    """PrioritizedBuffer over the entry ring: tree coordinate x == entry seq."""

    def __init__(self, store, capacity, max_size):
        from pfrl_amd.collections.prioritized import PrioritizedBuffer

        self.store = store
        self.capacity = capacity
        self.tree = PrioritizedBuffer(capacity=capacity, device=store.device, max_size=max_size)
        self._appends = 0

    def __len__(self):
        return len(self.tree)

    @property
    def head(self):
        return self.tree.frame.head

    def append_entry(self, tids):
        seq = self.store.add_entry(tids)
        self.tree.append(seq)
        assert self.tree.frame.next_x == self.store.n_entries
        self._appends += 1
        if (self._appends & 63) == 0 and self.store.frames is not None:
            oldest = self.store.h_e_min_fseq[self.head % self.store.E]
            if oldest < self.store.frames.oldest_live_seq():
                raise RuntimeError("frame ring too small for this replay capacity "
                                   "(n_slots=%d)" % self.store.frames.n_slots)

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

    @property
    def max_priority(self):
        return self.tree.max_priority


class PrioritizedReplayBuffer(ReplayBuffer, PriorityWeightError):
    """Stochastic prioritisation, proportional variant."""

    def __init__(self, capacity=None, alpha=0.6, beta0=0.4, betasteps=2e5, eps=0.01,
                 normalize_by_max=True, error_min=0, error_max=1, num_steps=1, device=None,
                 max_size=None, slack=None, frame_slots=None, priority_pow="device"):
        PriorityWeightError.__init__(self, alpha, beta0, betasteps, eps, normalize_by_max,
                                     error_min=error_min, error_max=error_max)
        # where np.float32 ** alpha of the priority transform (reference :47-55) is evaluated:
        #   "device"     one launch; glibc's powf restated on the device (csrc/powf_glibc.h), in
        #                the build (plain / FMA) this host's libm is probed to use: the leaves
        #                are bit for bit what NumPy computes here.  If the host's libm is not
        #                that powf, the buffer says so and evaluates on the host instead.
        #   "device_cr"  one launch; the correctly rounded power (rounds 1-2: <= 1 ulp from
        #                libm in ~0.05 % of inputs)
        #   "host_libm"  this host's libm through NumPy, one D2H per update
        assert priority_pow in ("device", "device_cr", "host_libm")
        self.priority_pow = priority_pow
        self._pow_mode = None
        ReplayBuffer.__init__(self, capacity=capacity, num_steps=num_steps, device=None,
                              max_size=max_size, slack=slack, frame_slots=frame_slots)
        if device is not None:
            self.bind(device)

    def _make_memory_host(self):
        # the gpu=None plumbing path: host trees over the caller's own scalars (bit-exact with
        # the reference by construction); the device path never comes here
        from pfrl_amd.collections.host_prioritized import HostPrioritizedBuffer

        return HostPrioritizedBuffer(capacity=self.capacity)

    def _make_memory_device(self):
        return _DevicePrioritizedQueue(self.store, self.capacity, self._device_opts["max_size"])

    def sample(self, n):
        self._ensure_bound()
        assert len(self.memory) >= n
        if self.store is None:
            # host lists of dicts: the weight rides on the first transition (reference :117-123)
            sampled, probabilities, min_prob = self.memory.sample(n)
            weights = self.weights_from_probabilities(probabilities, min_prob)
            for entry, w in zip(sampled, weights):
                entry[0]["weight"] = w
            return sampled
        tree = self.memory.tree
        # the store's new rows ride in the same host->device copy as the tree's pending leaf
        # writes and the uniform draws: one copy in front of the launches of this sample
        out = tree.sample_device(n, normalize=_NORMALIZE_CODE[self.normalize_by_max],
                                 beta=self.beta, slot_mod=self.store.E,
                                 co_stage=self.store.take_pending)
        self.beta = min(1.0, self.beta + self.beta_add)
        self._last_sample = out
        return DeviceExperienceBatch(self.store, out["slot"], _LazySeqs(out["x"], n),
                                     weights_dev=out["weight"])

    def sample_prepare(self, n):
        """``sample(n)`` in two parts for a caller that reaches the sample point before the
        previous minibatch's priorities exist (DQN._batch_observe_train_per): everything that
        does not depend on them happens now (host side, NumPy draws, the staging transfer, the
        store's new rows); the returned callable launches the rest and returns the batch."""
        self._ensure_bound()
        assert len(self.memory) >= n
        assert self.store is not None
        tree = self.memory.tree
        out, finish = tree.sample_device(n, normalize=_NORMALIZE_CODE[self.normalize_by_max],
                                         beta=self.beta, slot_mod=self.store.E,
                                         co_stage=self.store.take_pending, split=True)
        self.beta = min(1.0, self.beta + self.beta_add)

        def finish_sample():
            finish()
            self._last_sample = out
            return DeviceExperienceBatch(self.store, out["slot"], _LazySeqs(out["x"], n),
                                         weights_dev=out["weight"])

        return finish_sample

    def _native_state(self):
        tree = self.memory.tree
        tree.flush()
        torch.cuda.synchronize(self.device)
        f = tree.frame
        sd = dict(kind="pfrl_amd.PrioritizedReplayBuffer", capacity=self.capacity,
                  num_steps=self.num_steps, head=self.memory.head, beta=self.beta,
                  windows={k: list(v) for k, v in self.last_n_transitions.items()},
                  store=self.store.state_dict(self.memory.head),
                  tree=dict(sum_val=tree.sum_val.cpu(), sum_tag=tree.sum_tag.cpu(),
                            min_val=tree.min_val.cpu(), min_tag=tree.min_tag.cpu(),
                            maxp_val=tree._maxp_val.cpu(), maxp_tag=tree._maxp_tag.cpu(),
                            frame=dict(length=f.length, base=f.base, head=f.head,
                                       next_x=f.next_x, log2_size=f.log2_size,
                                       origin=list(f.origin), epoch=f.epoch),
                            data=list(tree.data)))
        return sd

    def _load_native(self, sd):
        assert sd["capacity"] == self.capacity and sd["num_steps"] == self.num_steps
        self.store.load_state_dict(sd["store"])
        self.beta = sd["beta"]
        tree = self.memory.tree
        t = sd["tree"]
        for name in ("sum_val", "sum_tag", "min_val", "min_tag"):
            getattr(tree, name).copy_(t[name].to(tree.device))
        tree._maxp_val.copy_(t["maxp_val"].to(tree.device))
        tree._maxp_tag.copy_(t["maxp_tag"].to(tree.device))
        for k, v in t["frame"].items():
            setattr(tree.frame, k, list(v) if k == "origin" else v)
        tree.data.clear()
        tree.data.extend(t["data"])
        tree.flag_wait_priority = False
        self.last_n_transitions.clear()
        for k, v in sd["windows"].items():
            self.last_n_transitions[k].extend(v)

    def update_errors(self, errors):
        """TD errors of the last sampled batch -> new priorities.

        ``errors`` may be a float32 device tensor (no host round trip), or a
        sequence of host scalars as in the reference."""
        if self.store is None:
            if isinstance(errors, torch.Tensor):
                errors = list(errors.detach().cpu().numpy().reshape(-1))
            self.memory.set_last_priority(self.priority_from_errors(errors))
            return
        tree = self.memory.tree
        if isinstance(errors, torch.Tensor):
            if self.priority_pow == "device" and self._pow_mode is None:
                from pfrl_amd import ops

                self._pow_mode = ops.powf_host_variant(self.alpha)
                if self._pow_mode is None:
                    import logging

                    logging.getLogger(__name__).warning(
                        "this host's powf is not glibc's: priorities are evaluated on the host")
                    self.priority_pow = "host_libm"
            if self.priority_pow in ("device", "device_cr") and errors.is_cuda:
                err = errors.detach().reshape(-1).to(torch.float32).contiguous()
                at_min = None if self.error_min is None else \
                    (self._clip(self.error_min) + self.eps) ** self.alpha
                at_max = None if self.error_max is None else \
                    (self._clip(self.error_max) + self.eps) ** self.alpha
                tree.update_errors_device(err, self.error_min, at_min, self.error_max, at_max,
                                          self.eps, self.alpha,
                                          pow_mode=self._pow_mode if self.priority_pow == "device" else 0)
                return
            # strict mode: evaluate the power with this host's libm, as NumPy does
            errors = list(errors.detach().cpu().numpy().reshape(-1))
        tree.set_last_priority(self.priority_from_errors(errors))
