"""Host + device profile of the bench step (run on the GPU box)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py", "--capacity", "100000", "--no-cpu-baseline"]
    args = bench.parse_args()
    device = torch.device("cuda:0")
    agent, env, rbuf = bench.build_agent(args, device, 0)
    N = args.num_envs
    obss = env.reset()
    obss = bench.prefill(agent, env, obss, N, 60000)
    for _ in range(3):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    for _ in range(3):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    pr.disable()
    print("3 steps: %.1f ms/step" % (el / 3 * 1e3))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            obss = bench.one_step(agent, env, obss, N)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))


if __name__ == "__main__":
    main()
