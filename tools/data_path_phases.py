"""Where one step of bench.py's ``data_path_only.without_update_launches`` spends its time:
host phases (act / env / observe / device tail) of the replay-side-only step, for reading next to a
``rocprofv3 --kernel-trace`` of the same command (tools/trace_slice.py on the last steps).
python tools/data_path_phases.py [--capacity 1000000]"""
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
obss = bench.prefill(agent, env, obss, N, args.capacity)
for _ in range(4):
    obss = bench.one_step(agent, env, obss, N)
out, obss = bench.data_path_only(args, dev, agent, env, rbuf, obss, 50)
print(out)
stub = bench.data_path_only.last_stub
torch.cuda.synchronize()
acc = np.zeros(5)
K = 100
for _ in range(K):
    t0 = time.perf_counter()
    actions = stub.batch_act(obss)
    t1 = time.perf_counter()
    obss, rs, dones, infos = env.step(actions)
    t2 = time.perf_counter()
    stub.batch_observe(obss, rs, dones, np.zeros(N, dtype=bool))
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    obss = env.reset(np.logical_not(dones))
    t5 = time.perf_counter()
    acc += (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)
print("per step (us), synchronised after observe: act %.1f  env %.1f  observe(host) %.1f  "
      "gpu tail %.1f  reset %.1f  total %.1f" % tuple(list(acc / K * 1e6) + [acc.sum() / K * 1e6]))
# free-running (what the bench figure is): only the act's action read-back synchronises
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    obss = bench.one_step(stub, env, obss, N)
torch.cuda.synchronize()
print("free-running: %.1f us/step" % ((time.perf_counter() - t0) / K * 1e6))
