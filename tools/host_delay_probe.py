"""Is a workload host-bound?  Adds a busy-wait of D microseconds to every update's host path and
reports throughput: a host-bound loop slows down by ~D per update, a device-bound one does not.
    python tools/host_delay_probe.py --algo rainbow"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
sys.argv = ["bench.py", "--no-cpu-baseline", "--capacity", "200000"] + sys.argv[1:]
args = bench.parse_args()
dev = torch.device("cuda:0")
agent, env, rbuf = bench.build_agent(args, dev, 0)
N = args.num_envs
obss = env.reset()
obss = bench.prefill(agent, env, obss, N, 60000 if args.algo != "sac" else 20000)
delay = [0.0]
name = "_update_from_batch" if hasattr(agent, "_update_from_batch") else "batch_observe"
per_call = 1 if name == "_update_from_batch" else N      # SAC: one update per env, delay x N per step
target = getattr(agent, name)
def slowed(*a, **k):
    t = time.perf_counter() + delay[0] * per_call
    r = target(*a, **k)
    while time.perf_counter() < t:
        pass
    return r
setattr(agent, name, slowed)
for _ in range(8):
    obss = bench.one_step(agent, env, obss, N)
for d in (0, 50, 100, 200, 0):
    delay[0] = d * 1e-6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(12):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("host delay %3d us per update: %.2f ms per step" % (d, el / 12 * 1e3), flush=True)
