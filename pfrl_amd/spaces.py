"""Minimal observation / action space descriptions, so that envs and wrappers work without gym.

They carry what the agents, wrappers and tests of this path read -- ``shape``, ``dtype``,
``low`` / ``high`` (Box), ``n`` (Discrete), ``sample()``, ``contains()`` -- and compare BY VALUE,
also against gym's own spaces (anything with the same attributes), because space objects cross
process boundaries in ``MultiprocessVectorEnv`` and are compared afterwards."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __eq__(self, other):
        return (hasattr(other, "low") and hasattr(other, "high")
                and tuple(getattr(other, "shape", ())) == self.shape
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    __hash__ = None

    def __repr__(self):
        return "Box({}, {}, {}, {})".format(self.low.min(), self.high.max(), self.shape, self.dtype)


class Discrete:
    def __init__(self, n, rng=None):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = rng        # private stream if given, else the global NumPy stream

    def sample(self):
        return int((self._rng or np.random).randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __eq__(self, other):
        return hasattr(other, "n") and not hasattr(other, "low") and int(other.n) == self.n

    __hash__ = None

    def __repr__(self):
        return "Discrete({})".format(self.n)
