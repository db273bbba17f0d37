"""A/B of env-range cuts inside ONE process (same box, same buffer, same clocks):
python tools/chunk_sweep.py "0.1,0.4" "0.125" "" "0.05,0.3" ..."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

configs = [a for a in sys.argv[1:]] or ["0.1,0.4", "0.125", ""]
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse_args()
dev = torch.device("cuda:0")
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.backends.cudnn.benchmark = True
agent, env, rbuf = bench.build_agent(args, dev, 0)
N = args.num_envs
obss = env.reset()
obss = bench.prefill(agent, env, obss, N, args.capacity)
for rep in range(3):
    for c in configs:
        agent.step_fused_chunks = tuple(float(x) for x in c.split(",") if x)
        for _ in range(6):
            obss = bench.one_step(agent, env, obss, N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            obss = bench.one_step(agent, env, obss, N)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("rep %d chunks %-12s %8.1f env-steps/s  %.3f ms/step" % (rep, repr(c), N * 40 / el, el / 40 * 1e3), flush=True)
