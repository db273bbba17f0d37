#!/usr/bin/env python
"""Rainbow on synthetic Atari-shaped envs -- the configuration of the reference's
examples/atari/reproduction/rainbow/train_rainbow.py (:110-159: CategoricalDoubleDQN,
DistributionalDuelingDQN with 51 atoms on [-10, 10], factorised NoisyNet sigma 0.5,
PrioritizedReplayBuffer(alpha 0.5, beta0 0.4, num_steps 3, normalize_by_max 'memory'),
Adam(6.25e-5, eps 1.5e-4), Greedy explorer) run through the batched driver with
`pfrl` replaced by `pfrl_amd`."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfrl_amd as pfrl  # noqa: E402
from pfrl_amd import agents, experiments, explorers, replay_buffers, utils  # noqa: E402
from pfrl_amd.q_functions import DistributionalDuelingDQN  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--num-envs", type=int, default=256)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--outdir", type=str, default="results")
    parser.add_argument("--steps", type=int, default=5 * 10 ** 7)
    parser.add_argument("--replay-start-size", type=int, default=2 * 10 ** 4)
    parser.add_argument("--capacity", type=int, default=10 ** 6)
    parser.add_argument("--eval-interval", type=int, default=250000)
    parser.add_argument("--eval-n-steps", type=int, default=None)
    parser.add_argument("--eval-n-runs", type=int, default=10)
    parser.add_argument("--n-best-episodes", type=int, default=200)
    args = parser.parse_args()

    import logging

    logging.basicConfig(level=logging.INFO)
    utils.set_random_seed(args.seed)
    os.makedirs(args.outdir, exist_ok=True)
    device = torch.device("cuda", args.gpu)
    N = args.num_envs

    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv

    def make_batch_env(test):
        slots = (args.capacity if not test else 0) + N * 32 + 8192
        store = DeviceFrameStore(slots, (84, 84), torch.uint8, device, stack=4)
        return SyntheticAtariVectorEnv(N, store=store, seed=args.seed + (10 ** 6 if test else 0))

    n_actions, n_atoms, v_max, v_min = 6, 51, 10, -10
    q_func = DistributionalDuelingDQN(n_actions, n_atoms, v_min, v_max)
    pfrl.nn.to_factorized_noisy(q_func, sigma_scale=0.5)   # noisy nets instead of an explorer
    q_func = q_func.to(memory_format=torch.channels_last)
    explorer = explorers.Greedy()
    opt = torch.optim.Adam(q_func.parameters(), 6.25e-5, eps=1.5 * 10 ** -4)
    update_interval = 4
    betasteps = args.steps / update_interval
    rbuf = replay_buffers.PrioritizedReplayBuffer(args.capacity, alpha=0.5, beta0=0.4,
                                                  betasteps=betasteps, num_steps=3,
                                                  normalize_by_max="memory")

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    agent = agents.CategoricalDoubleDQN(
        q_func, opt, rbuf, gpu=args.gpu, gamma=0.99, explorer=explorer, minibatch_size=32,
        replay_start_size=args.replay_start_size, target_update_interval=32000,
        update_interval=update_interval, batch_accumulator="mean", phi=phi)
    experiments.train_agent_batch_with_evaluation(
        agent=agent, env=make_batch_env(False), eval_env=make_batch_env(True), outdir=args.outdir,
        steps=args.steps, eval_n_steps=args.eval_n_steps, eval_n_episodes=args.eval_n_runs,
        eval_interval=args.eval_interval, log_interval=10 ** 5, save_best_so_far_agent=True)


if __name__ == "__main__":
    main()
