"""cProfile of the host side of the SAC bench step (python tools/host_profile_sac.py)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ["bench.py", "--algo", "sac", "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse_args()

dev = torch.device("cuda:0")
agent, env, rbuf = bench.build_agent(args, dev, 0)
obss = env.reset()
obss = bench.prefill(agent, env, obss, args.num_envs, args.prefill if args.prefill else 11000)
for _ in range(8):
    obss = bench.one_step(agent, env, obss, args.num_envs)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    obss = bench.one_step(agent, env, obss, args.num_envs)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
