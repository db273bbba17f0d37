"""Which torch ops launch what inside one (eager) DQN update?  torch.profiler over a
few updates; prints per-op device time and the memcpy / copy callers."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py", "--capacity", "100000", "--no-cpu-baseline", "--blas", "default"]
    args = bench.parse_args()
    device = torch.device("cuda", 0)
    agent, env, rbuf = bench.build_agent(args, device, 0)
    agent.use_graphs = False
    N = args.num_envs
    obss = env.reset()
    obss = bench.prefill(agent, env, obss, N, 60000)
    for _ in range(2):
        obss = bench.one_step(agent, env, obss, N)
    torch.cuda.synchronize()
    seqs = [rbuf.lookahead_sample(32) for _ in range(4)]
    big = rbuf.fetch_many(seqs, agent.phi, agent.gamma)
    ns = big["next_state"]
    raw = agent._precompute_target_raw(ns.view((4 * 32,) + tuple(ns.shape[2:])))
    big["target_next_raw"] = raw.view((4, 32) + tuple(raw.shape[1:]))
    for p in range(2):
        agent._update_from_batch({k: v[p] for k, v in big.items()})
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False,
                 record_shapes=True) as prof:
        for p in range(4):
            agent._update_from_batch({k: v[p] for k, v in big.items()})
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total",
                                                            row_limit=60, max_name_column_width=60))


if __name__ == "__main__":
    main()
