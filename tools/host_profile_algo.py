"""cProfile of the host side of one bench workload: python tools/host_profile_algo.py --algo rainbow"""
import cProfile, os, pstats, sys, time
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
if rbuf is not None:
    obss = bench.prefill(agent, env, obss, N, 60000 if args.algo != "sac" else 20000)
for _ in range(6):
    obss = bench.one_step(agent, env, obss, N)
torch.cuda.synchronize()
K = 10
from pfrl_amd.staging import StagingRing  # noqa: E402
import torch.cuda  # noqa: E402
_sync_wait = [0.0]
_orig_cpu = torch.Tensor.cpu
def _cpu(self, *a, **k):
    t = time.perf_counter(); r = _orig_cpu(self, *a, **k); _sync_wait[0] += time.perf_counter() - t; return r
torch.Tensor.cpu = _cpu
StagingRing.wait_s = 0.0
t0 = time.perf_counter()
for _ in range(K):
    obss = bench.one_step(agent, env, obss, N)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue time per step %.2f ms, wall per step %.2f ms; of the host time: %.2f ms waiting for "
      "staging slots, %.2f ms in .cpu() (D2H syncs)" % (t_host / K * 1e3, t_all / K * 1e3,
                                                       StagingRing.wait_s / K * 1e3, _sync_wait[0] / K * 1e3))
torch.Tensor.cpu = _orig_cpu
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    obss = bench.one_step(agent, env, obss, N)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
