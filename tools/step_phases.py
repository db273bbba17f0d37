"""Where one bench step spends its wall time: python tools/step_phases.py [--algo sac|dqn ...].
act = batch_act (ends with the D2H of the actions: waits for everything queued before),
env = env.step on the host, observe = batch_observe until it returns (launches queued),
tail = torch.cuda.synchronize() after observe (GPU work still outstanding)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse_args()
dev = torch.device("cuda:0")
agent, env, rbuf = bench.build_agent(args, dev, 0)
N = args.num_envs
obss = env.reset()
target = 10 ** 5 if args.algo == "sac" else min(args.capacity, 200000)
obss = bench.prefill(agent, env, obss, N, max(target, agent.replay_updater.replay_start_size + 1000))
for _ in range(8):
    obss = bench.one_step(agent, env, obss, N)
torch.cuda.synchronize()
acc = np.zeros(4)
K = 40
for _ in range(K):
    t0 = time.perf_counter()
    actions = agent.batch_act(obss)
    t1 = time.perf_counter()
    obss, rs, dones, infos = env.step(actions)
    t2 = time.perf_counter()
    agent.batch_observe(obss, rs, dones, np.zeros(N, dtype=bool))
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    obss = env.reset(np.logical_not(dones))
    acc += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
print("per step (ms): act %.3f  env %.3f  observe(host) %.3f  gpu tail %.3f  total %.3f"
      % tuple(list(acc / K * 1e3) + [acc.sum() / K * 1e3]))
