"""cProfile of the host side of one batched DQN step (data-path-only agent: zero-FLOP
q_function, so the GPU never holds the host up).  python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ["bench.py", "--capacity", "200000", "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse_args()
dev = torch.device("cuda:0")
agent, env, rbuf = bench.build_agent(args, dev, 0)
obss = env.reset()
obss = bench.prefill(agent, env, obss, args.num_envs, 60000)
from pfrl_amd import agents  # noqa: E402
from pfrl_amd.optimizers import FusedRMSprop  # noqa: E402

q = bench._ZeroFlopQ(6)
opt = FusedRMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
stub = agents.DQN(q, opt, rbuf, gpu=0, gamma=0.99, explorer=agent.explorer, replay_start_size=50000,
                  target_update_interval=30000, update_interval=4, minibatch_size=32,
                  batch_accumulator="sum", phi=agent.phi)
stub.t = agent.t
if os.environ.get("FULL") == "1":
    stub = agent      # the real network: where the host spends its time when the GPU is busy
for _ in range(5):
    obss = bench.one_step(stub, env, obss, args.num_envs)
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    obss = bench.one_step(stub, env, obss, args.num_envs)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
