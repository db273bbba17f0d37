"""PER tree kernels at capacity 1e6 (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfrl_amd.collections.prioritized import PrioritizedBuffer  # noqa: E402


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = torch.device("cuda:0")
    buf = PrioritizedBuffer(cap, device=dev)
    rs = np.random.RandomState(0)
    t0 = time.perf_counter()
    for i in range(cap + 5000):
        buf.append(i)
        if (i & 1023) == 1023:
            buf.flush()
    buf.flush()
    torch.cuda.synchronize()
    print("fill %.1f s" % (time.perf_counter() - t0))
    # realistic spread of priorities
    for _ in range(200):
        out = buf.sample_device(1024, u01=rs.random_sample(1024))
        err = torch.from_numpy((rs.rand(1024) * 1.2).astype(np.float32)).to(dev)
        buf.update_errors_device(err, 0, 0.01 ** 0.5, 1, 1.01 ** 0.5, 0.01, 0.5)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tot = np.zeros(3)
    for it in range(iters):
        for j in range(4):
            buf.append(cap + 10000 + it * 4 + j)
        ev[0].record()
        buf.flush()
        ev[1].record()
        out = buf.sample_device(32, u01=rs.random_sample(32), normalize=2, beta=0.5, slot_mod=cap)
        ev[2].record()
        err = torch.from_numpy((rs.rand(32) * 1.2).astype(np.float32)).to(dev)
        buf.update_errors_device(err, 0, 0.01 ** 0.5, 1, 1.01 ** 0.5, 0.01, 0.5)
        ev[3].record()
        torch.cuda.synchronize()
        tot += [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])]
    print("per update (events, us): write %.1f  sample %.1f  update %.1f"
          % tuple(tot / iters * 1e3))


if __name__ == "__main__":
    main()
