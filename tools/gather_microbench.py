"""Fused gather (pfrl_batch_experiences) at the bench's launch size: 2048
entries whose state / next_state stacks share 3 of 4 frames."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfrl_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    F, R, B, k = 300000, 250000, int(os.environ.get("B", 2048)), 4
    frames = torch.randint(0, 256, (F, 84, 84), dtype=torch.uint8, device=dev)
    base = rs.randint(0, F - 8, size=R)
    state = np.stack([base + j for j in range(k)], axis=1).astype(np.int32)
    nxt = state + 1
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_state, d_next = T(state), T(nxt.astype(np.int32))
    d_act = torch.zeros(R, dtype=torch.int64, device=dev)
    d_rew = torch.zeros(R, dtype=torch.float64, device=dev)
    d_term = torch.zeros(R, dtype=torch.uint8, device=dev)
    d_tids = T(np.arange(R, dtype=np.int32).reshape(R, 1))
    d_len = torch.ones(R, dtype=torch.int32, device=dev)
    desc = ops.make_table_desc(d_state, d_next, d_act, d_rew, d_term, d_tids, d_len, k, 1, 0)
    out = dict(state=torch.empty((B, k, 84, 84), dtype=torch.float32, device=dev),
               next_state=torch.empty((B, k, 84, 84), dtype=torch.float32, device=dev),
               action=torch.empty(B, dtype=torch.int64, device=dev),
               reward=torch.empty(B, dtype=torch.float32, device=dev),
               is_state_terminal=torch.empty(B, dtype=torch.float32, device=dev),
               discount=torch.empty(B, dtype=torch.float32, device=dev))
    slots = [T(rs.randint(0, R, size=B).astype(np.int32)) for _ in range(8)]
    for s in slots:
        ops.batch_experiences(desc, frames, 255.0, s, [1.0, 0.99], out)
    torch.cuda.synchronize()
    n = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.batch_experiences(desc, frames, 255.0, slots[i % 8], [1.0, 0.99], out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    alg = B * 2 * k * 5 * 7056
    print("B=%d  %.1f us/launch  %.2f TB/s algorithmic (%.3f of 8 TB/s)" % (B, us, alg / us / 1e6,
                                                                              alg / us / 1e6 / 8))
    # correctness of the shared-frame path
    ref_s = frames[d_state[slots[(n - 1) % 8].long()].long()]
    ref_n = frames[d_next[slots[(n - 1) % 8].long()].long()]
    back = lambda t: (t * 255).round().to(torch.uint8)
    assert torch.equal(back(out["state"]), ref_s) and torch.equal(back(out["next_state"]), ref_n)
    print("check ok")


if __name__ == "__main__":
    main()
