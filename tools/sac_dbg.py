"""Where does a SAC step go?  (host segments timed with a device sync after each)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py", "--algo", "sac"]
    args = bench.parse_args()
    device = torch.device("cuda", 0)
    agent, env, rbuf = bench.build_agent(args, device, 0)
    N = args.num_envs
    obss = env.reset()
    obss = bench.prefill(agent, env, obss, N, 20000)
    for _ in range(3):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()

    def timeit(name, fn, n=50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        print("%-28s %8.1f us" % (name, (time.perf_counter() - t0) / n * 1e6))
        return r

    from pfrl_amd.replay_buffer import batch_experiences

    exps = timeit("rbuf.sample(256)", lambda: rbuf.sample(256))
    batch = timeit("batch_experiences", lambda: batch_experiences(exps, device, agent.phi, 0.99))
    tensors = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    if agent._captured is not None:
        timeit("graph replay", lambda: agent._captured.run(tensors))
    timeit("agent.update(exps)", lambda: agent.update(exps))
    timeit("update_if_necessary", lambda: agent.replay_updater.update_if_necessary(agent.t))
    acts = timeit("batch_act", lambda: agent.batch_act(obss), n=10)
    timeit("env.step", lambda: env.step(acts), n=10)
    saved = agent.replay_updater.replay_start_size
    agent.replay_updater.replay_start_size = 1 << 62
    o2, r, d, _ = env.step(acts)
    timeit("batch_observe (no updates)",
           lambda: agent.batch_observe(o2, r, d, np.zeros(N, dtype=bool)), n=10)
    agent.replay_updater.replay_start_size = saved


if __name__ == "__main__":
    main()
