"""Integer bookkeeping of the reference's TreeQueue, device-agnostic.

The reference keeps a pointer tree over a sliding index frame
``bounds = (ixl, ixr)`` that doubles on append and halves / re-roots on
popleft (pfrl/collections/prioritized.py:207-242).  Elements never move inside
that frame, so every element can be given an *absolute* leaf coordinate
``x`` (its append sequence number); the frame is then ``[base, base + 2**L)``
and logical index ``i`` is ``x = head + i``.  This class tracks
(base, head, length, L) and the per-level alignment origins exactly as the
reference's bounds evolve; the device kernels receive them in ``pfrl_tree_t``.

``epoch`` increases whenever the frame changes: pending leaf writes recorded
under the previous frame must be flushed to the device first, because their
ancestor repair has to run in the frame they were issued in.
"""
from pfrl_amd._native import MAX_LEVELS


class TreeFrame:
    def __init__(self):
        self.length = 0
        self.base = 0
        self.head = 0
        self.next_x = 0
        self.log2_size = 0
        self.origin = [0] * MAX_LEVELS
        self.epoch = 0

    @property
    def size(self):
        return 1 << self.log2_size

    @property
    def bounds(self):
        """The reference's ``TreeQueue.bounds`` (relative to logical index 0)."""
        ixl = self.base - self.head
        return ixl, ixl + self.size

    def will_change_on_append(self):
        return self.length == 0 or self.next_x == self.base + self.size

    def append(self):
        """prioritized.py:207-223.  Returns the new element's coordinate."""
        if self.length == 0:
            self.base = self.head = self.next_x
            self.log2_size = 0
            self.epoch += 1
        elif self.next_x == self.base + self.size:  # ixr == length
            self.log2_size += 1
            self.origin[self.log2_size] = self.base
            self.epoch += 1
        x = self.next_x
        self.next_x += 1
        self.length += 1
        return x

    def popleft_coord(self):
        return self.head

    def will_change_on_popleft(self):
        if self.length == 1:
            return True
        return self.head + 1 == self.base + self.size // 2

    def popleft(self):
        """prioritized.py:225-242.  Returns the removed element's coordinate.
        The caller must have flushed the leaf deletion under the old frame when
        ``will_change_on_popleft()`` was true."""
        assert self.length > 0
        x = self.head
        self.head += 1
        self.length -= 1
        if self.length == 0:
            self.epoch += 1
        elif self.log2_size > 0 and self.head == self.base + self.size // 2:  # ixc == 0
            self.base += self.size // 2
            self.log2_size -= 1
            self.epoch += 1
        return x


def appends_keep_frame(frame, m, pops):
    """True when the next ``m`` appends -- each preceded by a ``popleft`` if ``pops`` (a buffer
    at capacity) -- leave ``frame`` (a :class:`TreeFrame`) as it is: no doubling
    (prioritized.py:207-223), no halving / re-rooting (:225-242), i.e. ``epoch`` stays.
    Conservative: may say False for a sequence that would in fact keep the frame."""
    if frame.length <= m + 1:
        return False
    if frame.next_x + m > frame.base + frame.size:
        return False
    if pops and frame.head + m >= frame.base + frame.size // 2:
        return False
    return True


def smax_log2_for_capacity(capacity):
    """Largest frame the reference can reach with this capacity: doubling needs
    more than size/2 live items, so size <= 2 * next_pow2(capacity)."""
    L = 0
    while (1 << L) < max(int(capacity), 1):
        L += 1
    return L + 1
