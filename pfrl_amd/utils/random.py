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
    # Positions whose value already occurs earlier among the first k (stable sort: equal values
    # keep their index order, so the earliest occurrence of a value owns it).  Usually none
    # (k^2 / 2n ~ 5e-4 for 32 of 10^6): the repair walk would change and consume nothing.
    srt = np.sort(head)
    if not (srt[1:] == srt[:-1]).any():
        return head
    order = np.argsort(head, kind="stable")
    srt = head[order]
    same = srt[1:] == srt[:-1]
    # The reference walks i = 0 .. k-1 with a set of the values kept so far and replaces a
    # value found in the set by the next spare (draws[k:], refilled when used up).  Only the
    # duplicate positions ever change, so only those are visited here, in index order; a spare
    # is rejected exactly when an EARLIER position holds its value at that moment, and a
    # spare equal to the value of a LATER position takes it over (that position is then a
    # duplicate in turn, as it would be found when the walk reaches it).
    import bisect

    pending = sorted(order[1:][same].tolist())
    owner = {}                 # values whose owning position changed: value -> position
    spare = k
    while pending:
        d = pending.pop(0)
        while True:
            x = draws[spare]
            spare += 1
            if spare == 2 * k:
                # (the refill draws happen at the same point of the stream as in the walk)
                draws[k:] = np.random.randint(0, n, size=k)
                spare = k
            xi = int(x)
            pos = owner.get(xi)
            if pos is None:
                j = np.searchsorted(srt, x)
                if j < k and srt[j] == x:
                    cand = int(order[j])           # earliest original holder of x ...
                    # ... unless that position was itself a duplicate already replaced
                    pos = cand if head[cand] == x else None
                    if pos is None:
                        # look for a later original holder that still has x
                        jj = j + 1
                        while jj < k and srt[jj] == x:
                            c2 = int(order[jj])
                            if head[c2] == x:
                                pos = c2
                                break
                            jj += 1
            if pos is not None and pos < d:
                continue                            # in the set of kept values: next spare
            head[d] = x
            owner[xi] = d
            if pos is not None:                     # pos > d: that position is now a duplicate
                bisect.insort(pending, pos)
            break
    return head
