"""Host-side prioritized buffer for the ``gpu=None`` plumbing path
(reference pfrl/collections/prioritized.py:21-323).

The device path keeps the sum / min trees in HBM (``collections/prioritized.py``, HIP kernels) and
never comes here; this module serves buffers that are used without a GPU, with the reference's
exact results: priorities are kept as the Python / NumPy scalars the caller passed in and every
node is ``sum`` / ``min`` over the list of its present children, so NEP-50 type promotion, the
order of additions and "absent" subtrees behave as in the reference by construction.

Layout.  The reference's pointer tree over a sliding, re-rooting index frame becomes one dict per
level, keyed by node number within that level: level 0 holds the leaves at their absolute
coordinate ``x`` (append sequence number), a node ``q`` of level ``l`` covers the coordinates
``[origin[l] + q * 2**l, origin[l] + (q + 1) * 2**l)``.  ``TreeFrame`` (shared with the device
implementation) tracks the frame ``[base, base + 2**L)``, the per-level origins and when the frame
doubles (append past its right edge) or halves (left half empty after ``popleft``).  A missing key
is an absent node; a node whose children are both absent is removed.
"""
import collections

import numpy as np

from pfrl_amd._native import MAX_LEVELS
from pfrl_amd.collections.tree_frame import TreeFrame
from pfrl_amd.utils.random import sample_n_k


class _TreeQueue:
    """FIFO of scalars with an ``op``-reduction cached over it (``TreeQueue``, reference :180-242)."""

    def __init__(self, op):
        self.op = op
        self.frame = TreeFrame()
        self.levels = [{} for _ in range(MAX_LEVELS)]

    @property
    def length(self):
        return self.frame.length

    @property
    def bounds(self):
        return self.frame.bounds

    def _children(self, l, q):
        """Node numbers at level ``l - 1`` of the children of node ``q`` at level ``l``."""
        f = self.frame
        start = f.origin[l] + (q << l)
        left = (start - f.origin[l - 1]) >> (l - 1)
        return left, left + 1

    def _root(self):
        f = self.frame
        return self.levels[f.log2_size][(f.base - f.origin[f.log2_size]) >> f.log2_size]

    def _write_x(self, x, value):
        """Set (``value is None``: remove) the leaf at coordinate ``x`` and repair its ancestors
        up to the current root.  Returns the previous leaf value or None (reference :153-177)."""
        f = self.frame
        leaves = self.levels[0]
        previous = leaves.get(x)
        if value is None:
            leaves.pop(x, None)
        else:
            leaves[x] = value
        for l in range(1, f.log2_size + 1):
            q = (x - f.origin[l]) >> l
            below = self.levels[l - 1]
            present = [below[c] for c in self._children(l, q) if c in below]
            if present:
                self.levels[l][q] = self.op(present)
            else:
                self.levels[l].pop(q, None)
        return previous

    def _write(self, ix, value):
        return self._write_x(self.frame.head + ix, value)

    def __setitem__(self, ix, value):
        assert 0 <= ix < self.length
        assert value is not None
        self._write(ix, value)

    def append(self, value):
        f = self.frame
        if f.length and f.will_change_on_append():
            self.levels[f.log2_size + 1].clear()       # the new root level starts empty
        x = f.append()
        previous = self._write_x(x, value)
        assert previous is None

    def popleft(self):
        f = self.frame
        assert f.length > 0
        previous = self._write_x(f.popleft_coord(), None)
        top = f.log2_size
        f.popleft()
        if f.length and f.log2_size < top:              # re-rooted at the old right child
            self.levels[top].clear()
        return previous


class _SumTreeQueue(_TreeQueue):
    def __init__(self):
        super().__init__(op=sum)

    def sum(self):
        return 0.0 if self.length == 0 else self._root()

    def _find(self, pos):
        """Logical index of the leaf at cumulative position ``pos`` (reference :245-258)."""
        f = self.frame
        l = f.log2_size
        q = (f.base - f.origin[l]) >> l
        while l > 0:
            left, right = self._children(l, q)
            left_value = self.levels[l - 1].get(left, 0.0)
            if pos < left_value:
                q = left
            else:
                pos = pos - left_value
                q = right
            l -= 1
        return q - f.head

    def _take(self, ixs, remove):
        vals = []
        for ix in ixs:
            val = self._write(ix, 0.0)
            assert val is not None
            vals.append(val)
        if not remove:
            for ix, val in zip(ixs, vals):
                self._write(ix, val)
        return ixs, vals

    def uniform_sample(self, n, remove):
        assert n >= 0
        return self._take(list(sample_n_k(self.length, n)), remove)

    def prioritized_sample(self, n, remove):
        """``n`` sequential draws; each drawn leaf is zeroed before the next draw, so the same
        item is not drawn twice and the total shrinks as the reference's does (:294-312)."""
        assert n >= 0
        ixs, vals = [], []
        for _ in range(n):
            ix = self._find(np.random.uniform(0.0, self._root()))
            val = self._write(ix, 0.0)
            assert val is not None
            ixs.append(ix)
            vals.append(val)
        if not remove:
            for ix, val in zip(ixs, vals):
                self._write(ix, val)
        return ixs, vals


class _MinTreeQueue(_TreeQueue):
    def __init__(self):
        super().__init__(op=min)

    def min(self):
        return np.inf if self.length == 0 else self._root()


class HostPrioritizedBuffer:
    """``PrioritizedBuffer`` on the host (reference :21-123): same attributes, same assertions."""

    def __init__(self, capacity=None, wait_priority_after_sampling=True,
                 initial_max_priority=1.0):
        self.capacity = capacity
        self.data = collections.deque()
        self.priority_sums = _SumTreeQueue()
        self.priority_mins = _MinTreeQueue()
        self.max_priority = initial_max_priority
        self.wait_priority_after_sampling = wait_priority_after_sampling
        self.flag_wait_priority = False

    def __len__(self):
        return len(self.data)

    def append(self, value, priority=None):
        if self.capacity is not None and len(self) == self.capacity:
            self.popleft()
        if priority is None:
            priority = self.max_priority
        self.data.append(value)
        self.priority_sums.append(priority)
        self.priority_mins.append(priority)

    def popleft(self):
        assert len(self) > 0
        self.priority_sums.popleft()
        self.priority_mins.popleft()
        return self.data.popleft()

    def _sample_indices_and_probabilities(self, n, uniform_ratio):
        total = self.priority_sums.sum()
        min_prob = self.priority_mins.min() / total
        indices, priorities = [], []
        remove = self.wait_priority_after_sampling
        if uniform_ratio > 0:
            n_uniform = np.random.binomial(n, uniform_ratio)
            ixs, pris = self.priority_sums.uniform_sample(n_uniform, remove=remove)
            indices.extend(ixs)
            priorities.extend(pris)
            n -= n_uniform
            min_prob = uniform_ratio / len(self) + (1 - uniform_ratio) * min_prob
        ixs, pris = self.priority_sums.prioritized_sample(n, remove=remove)
        indices.extend(ixs)
        priorities.extend(pris)
        probs = [uniform_ratio / len(self) + (1 - uniform_ratio) * pri / total
                 for pri in priorities]
        return indices, probs, min_prob

    def sample(self, n, uniform_ratio=0):
        assert not self.wait_priority_after_sampling or not self.flag_wait_priority
        indices, probabilities, min_prob = self._sample_indices_and_probabilities(
            n, uniform_ratio=uniform_ratio)
        sampled = [self.data[i] for i in indices]
        self.sampled_indices = indices
        self.flag_wait_priority = True
        return sampled, probabilities, min_prob

    def set_last_priority(self, priority):
        assert not self.wait_priority_after_sampling or self.flag_wait_priority
        assert all([p > 0.0 for p in priority])
        assert len(self.sampled_indices) == len(priority)
        for i, p in zip(self.sampled_indices, priority):
            self.priority_sums[i] = p
            self.priority_mins[i] = p
            self.max_priority = max(self.max_priority, p)
        self.flag_wait_priority = False
        self.sampled_indices = []
