"""Host-side timeline of one DQN bench step (no extra device syncs): how long the
GPU waits for the host between the acting forward pass and the first update."""
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

ACC = defaultdict(float)


def main():
    sys.argv = ["bench.py", "--capacity", "200000", "--no-cpu-baseline"] + sys.argv[1:]
    args = bench.parse_args()
    device = torch.device("cuda", 0)
    agent, env, rbuf = bench.build_agent(args, device, 0)
    N = args.num_envs
    obss = env.reset()
    obss = bench.prefill(agent, env, obss, N, 100000)
    for _ in range(4):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()

    # instrument the internals of the fused observe path
    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def w(*a, **kw):
            t0 = time.perf_counter()
            r = fn(*a, **kw)
            ACC[label] += time.perf_counter() - t0
            return r

        setattr(obj, name, w)

    wrap(rbuf, "lookahead_sample", "  observe: lookahead_sample (64x)")
    wrap(rbuf, "fetch_many", "  observe: fetch_many (gather launch)")
    wrap(rbuf, "append", "  observe: append (256x)")
    wrap(agent, "_update_from_batch", "  observe: _update_from_batch (64x, enqueue)")
    if hasattr(agent, "_precompute_target_raw"):
        wrap(agent, "_precompute_target_raw", "  observe: target pass (enqueue)")
    steps = 20
    t_all = time.perf_counter()
    for _ in range(steps):
        t0 = time.perf_counter()
        actions = agent.batch_act(obss)
        t1 = time.perf_counter()
        obss2, rs, dones, infos = env.step(actions)
        t2 = time.perf_counter()
        agent.batch_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
        t3 = time.perf_counter()
        obss = env.reset(np.logical_not(dones))
        t4 = time.perf_counter()
        ACC["batch_act (incl. wait for previous step's GPU work + D2H)"] += t1 - t0
        ACC["env.step"] += t2 - t1
        ACC["batch_observe (host enqueue time)"] += t3 - t2
        ACC["env.reset"] += t4 - t3
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    print("wall per step %.2f ms" % (wall / steps * 1e3))
    for k, v in ACC.items():
        print("%-62s %8.2f ms/step" % (k, v / steps * 1e3))


if __name__ == "__main__":
    main()
