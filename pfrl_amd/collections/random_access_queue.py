"""Host FIFO with O(1) indexing, used by the CPU (gpu=-1) replay path.

Same behaviour as ``pfrl.collections.random_access_queue.RandomAccessQueue``
(/root/reference/pfrl/collections/random_access_queue.py:6-101): append,
popleft, indexing from either end, optional ``maxlen`` eviction, ``sample``.
Implemented as one growing list plus a head offset that is compacted lazily.
"""
import itertools

from pfrl_amd.utils.random import sample_n_k


class RandomAccessQueue(object):
    def __init__(self, *args, **kwargs):
        self.maxlen = kwargs.pop("maxlen", None)
        assert self.maxlen is None or self.maxlen >= 0
        self._items = list(*args, **kwargs)
        self._head = 0
        self._apply_maxlen()

    def __setstate__(self, state):
        """Also accepts the attribute dict of the reference's class (two lists,
        ``_queue_front`` reversed + ``_queue_back``; random_access_queue.py:13-17), so that a
        queue pickled by the reference loads into this one."""
        if "_queue_back" in state:
            self.maxlen = state.get("maxlen")
            self._items = list(reversed(state["_queue_front"])) + list(state["_queue_back"])
            self._head = 0
        else:
            self.__dict__.update(state)

    def _apply_maxlen(self):
        if self.maxlen is not None:
            while len(self) > self.maxlen:
                self.popleft()

    def _compact(self):
        if self._head > 1024 and self._head * 2 > len(self._items):
            del self._items[: self._head]
            self._head = 0

    def __len__(self):
        return len(self._items) - self._head

    def __iter__(self):
        return itertools.islice(self._items, self._head, None)

    def __repr__(self):
        return "RandomAccessQueue({})".format(str(list(iter(self))))

    def _pos(self, i):
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("RandomAccessQueue index out of range")
        return self._head + i

    def __getitem__(self, i):
        return self._items[self._pos(i)]

    def __setitem__(self, i, x):
        self._items[self._pos(i)] = x

    def append(self, x):
        self._items.append(x)
        if self.maxlen is not None and len(self) > self.maxlen:
            self.popleft()

    def extend(self, xs):
        self._items.extend(xs)
        self._apply_maxlen()

    def popleft(self):
        if len(self) == 0:
            raise IndexError("pop from empty RandomAccessQueue")
        x = self._items[self._head]
        self._items[self._head] = None
        self._head += 1
        self._compact()
        return x

    def sample(self, k):
        return [self[i] for i in sample_n_k(len(self), k)]
