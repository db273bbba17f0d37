"""Distinct uniform sampling on the legacy global NumPy stream.

Restates ``pfrl.utils.random.sample_n_k`` (/root/reference/pfrl/utils/random.py:
4-28).  Parity needs the *same consumption* of the global ``np.random`` stream,
so the draws stay on the host (they are a few dozen integers); only the
resulting indices travel to the device.

Two regimes, as in the reference.  Dense (3k >= n): one ``choice(n, k, replace=False)``, i.e. a
full permutation draw.  Sparse: draw 2k candidates with replacement, keep the first k, and repair
duplicates among them from the spare half in order; if the spares run out (rare), redraw k spares.
For replay sampling (k = 32 out of 10^6) the sparse branch is one 64-integer draw and a set
walk.  ``tests/golden/sample_n_k.npz`` pins both the indices and the stream position afterwards
(the next draw of the global stream), since everything downstream shares that stream.
"""
import numpy as np


def sample_n_k(n, k):
    """Sample k distinct elements uniformly from range(n)."""
    if not 0 <= k <= n:
        raise ValueError("Sample larger than population or is negative")
    if k == 0:
        return np.empty((0,), dtype=np.int64)
    if 3 * k >= n:
        return np.random.choice(n, k, replace=False)
    # RandomState.choice(n, size) with replacement is randint(0, n, size) on
    # the same stream; calling randint directly skips choice()'s argument
    # checking (the dominant cost for k = 32).
    draws = np.random.randint(0, n, size=2 * k)
    head = draws[:k]
    # no duplicate among the first k (the usual case: k^2 / 2n ~ 5e-4 for 32 of 10^6): the
    # repair walk below would change nothing and consume nothing more
    srt = np.sort(head)
    if not (srt[1:] == srt[:-1]).any():
        return head
    seen = set()
    spare = k
    for i in range(k):
        x = draws[i]
        while x in seen:
            x = draws[i] = draws[spare]
            spare += 1
            if spare == 2 * k:
                draws[k:] = np.random.randint(0, n, size=k)
                spare = k
        seen.add(x)
    return draws[:k]
