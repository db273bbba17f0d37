#!/usr/bin/env python
"""Batched DQN on synthetic Atari-shaped envs -- the structure of the
reference's examples/atari/train_dqn_batch_ale.py with `pfrl` replaced by
`pfrl_amd` and the ALE/gym env factory replaced by the on-device synthetic
VectorEnv (there is no ALE in this image).  Everything from `q_func` down is
the reference script's code."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfrl_amd as pfrl  # noqa: E402
from pfrl_amd import agents, experiments, explorers  # noqa: E402
from pfrl_amd import nn as pnn  # noqa: E402
from pfrl_amd import replay_buffers, utils  # noqa: E402
from pfrl_amd.initializers import init_chainer_default  # noqa: E402
from pfrl_amd.q_functions import DiscreteActionValueHead, DuelingDQN  # noqa: E402


def parse_arch(arch, n_actions):
    if arch == "nature":
        return nn.Sequential(pnn.LargeAtariCNN(), init_chainer_default(nn.Linear(512, n_actions)),
                             DiscreteActionValueHead())
    if arch == "nips":
        return nn.Sequential(pnn.SmallAtariCNN(), init_chainer_default(nn.Linear(256, n_actions)),
                             DiscreteActionValueHead())
    if arch == "dueling":
        return DuelingDQN(n_actions)
    raise RuntimeError("Not supported architecture: {}".format(arch))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--outdir", type=str, default="results")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--final-exploration-frames", type=int, default=10 ** 6)
    parser.add_argument("--final-epsilon", type=float, default=0.01)
    parser.add_argument("--arch", type=str, default="nature", choices=["nature", "nips", "dueling"])
    parser.add_argument("--steps", type=int, default=10 ** 6)
    parser.add_argument("--replay-start-size", type=int, default=5 * 10 ** 4)
    parser.add_argument("--target-update-interval", type=int, default=3 * 10 ** 4)
    parser.add_argument("--eval-interval", type=int, default=10 ** 5)
    parser.add_argument("--update-interval", type=int, default=4)
    parser.add_argument("--eval-n-runs", type=int, default=10)
    parser.add_argument("--agent", type=str, default="DoubleDQN", choices=["DQN", "DoubleDQN"])
    parser.add_argument("--prioritized", action="store_true", default=False)
    parser.add_argument("--lr", type=float, default=2.5e-4)
    parser.add_argument("--num-envs", type=int, default=256)
    parser.add_argument("--n-step-return", type=int, default=1)
    parser.add_argument("--capacity", type=int, default=10 ** 6)
    args = parser.parse_args()

    import logging

    logging.basicConfig(level=logging.INFO)
    utils.set_random_seed(args.seed)
    os.makedirs(args.outdir, exist_ok=True)
    device = torch.device("cuda", args.gpu)

    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv

    def make_batch_env(test):
        slots = (args.capacity if not test else 0) + args.num_envs * 32 + 8192
        store = DeviceFrameStore(slots, (84, 84), torch.uint8, device, stack=4)
        return SyntheticAtariVectorEnv(args.num_envs, store=store,
                                       seed=args.seed + (10 ** 6 if test else 0))

    n_actions = 6
    q_func = parse_arch(args.arch, n_actions)
    opt = torch.optim.RMSprop(q_func.parameters(), lr=args.lr, alpha=0.95, momentum=0.0, eps=1e-2,
                              centered=True)
    if args.prioritized:
        betasteps = args.steps / args.update_interval
        rbuf = replay_buffers.PrioritizedReplayBuffer(args.capacity, alpha=0.6, beta0=0.4,
                                                      betasteps=betasteps,
                                                      num_steps=args.n_step_return)
    else:
        rbuf = replay_buffers.ReplayBuffer(args.capacity, num_steps=args.n_step_return)
    explorer = explorers.LinearDecayEpsilonGreedy(
        1.0, args.final_epsilon, args.final_exploration_frames,
        lambda: np.random.randint(n_actions))

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    Agent = {"DQN": agents.DQN, "DoubleDQN": agents.DoubleDQN}[args.agent]
    agent = Agent(q_func, opt, rbuf, gpu=args.gpu, gamma=0.99, explorer=explorer,
                  replay_start_size=args.replay_start_size,
                  target_update_interval=args.target_update_interval, clip_delta=True,
                  update_interval=args.update_interval, batch_accumulator="sum", phi=phi)
    experiments.train_agent_batch_with_evaluation(
        agent=agent, env=make_batch_env(test=False), eval_env=make_batch_env(test=True),
        steps=args.steps, eval_n_steps=None, eval_n_episodes=args.eval_n_runs,
        eval_interval=args.eval_interval, outdir=args.outdir, save_best_so_far_agent=False,
        log_interval=10 ** 4)
    print(agent.get_statistics())


if __name__ == "__main__":
    main()
