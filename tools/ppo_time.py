"""Where does a PPO step go?  Rollout-phase segments and update-phase segments,
each bracketed by a device sync (so numbers are serialised, i.e. upper bounds)."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

ACC = defaultdict(float)
CNT = defaultdict(int)


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrapper(*a, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **kw)
        torch.cuda.synchronize()
        ACC[label] += time.perf_counter() - t0
        CNT[label] += 1
        return r

    setattr(obj, name, wrapper)


def main():
    sys.argv = ["bench.py", "--algo", "ppo"] + sys.argv[1:]
    args = bench.parse_args()
    device = torch.device("cuda", 0)
    agent, env, _ = bench.build_agent(args, device, 0)
    N = args.num_envs
    obss = env.reset()
    for _ in range(128):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()
    for name in ("batch_act", "batch_observe", "_update", "_value_pass", "_gather", "_lossfun"):
        if hasattr(agent, name):
            timed(agent, name)
    timed(env, "step", "env.step")
    timed(env, "reset", "env.reset")
    timed(agent.optimizer, "step", "optimizer.step")
    t0 = time.perf_counter()
    for _ in range(128):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("wall %.1f ms for 128 steps (%.0f env-steps/s, serialised by the timers)"
          % (wall * 1e3, 128 * N / wall))
    for k in sorted(ACC, key=lambda k: -ACC[k]):
        print("%-16s calls %5d  total %8.1f ms  avg %8.1f us" % (k, CNT[k], ACC[k] * 1e3,
                                                               ACC[k] / CNT[k] * 1e6))


if __name__ == "__main__":
    main()
