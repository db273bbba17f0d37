"""Host + device profile of the bench step (run on the GPU box)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py", "--capacity", "100000", "--no-cpu-baseline", "--no-cudnn-benchmark"]
    args = bench.parse_args()
    device = torch.device("cuda:0")
    agent, env, rbuf = bench.build_agent(args, device, 0)
    N = args.num_envs
    obss = env.reset()
    obss = bench.prefill(agent, env, obss, N, 60000)
    for _ in range(4):
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
    print(s.getvalue()[:7000])
    # host-only time per step (no sync inside): how far ahead of the GPU the host runs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        obss = bench.one_step(agent, env, obss, N)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("5 steps: host enqueue %.1f ms/step, wall %.1f ms/step" % (host / 5 * 1e3, tot / 5 * 1e3))


if __name__ == "__main__":
    main()
