#!/usr/bin/env python
"""Recurrent DQN (DRQN) on synthetic Atari-shaped envs -- the agent side of the reference's
examples/atari/train_drqn_ale.py with `pfrl` replaced by `pfrl_amd`: a CNN + LSTM Q-function in
a ``RecurrentSequential``, an ``EpisodicReplayBuffer`` and ``DoubleDQN(recurrent=True)`` that
replays sub-episodes of ``--episodic-update-len`` steps.  Episodes live on the host and the
update runs eagerly (packed sequences of varying shape), so this also runs with ``--gpu -1``.

    python examples/train_drqn_batch_synthetic.py --gpu -1 --steps 2000 --num-envs 8 \\
        --p-done 0.05 --replay-start-size 300 --batch-size 4
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfrl_amd as pfrl  # noqa: E402
from pfrl_amd import agents, experiments, explorers, replay_buffers, utils  # noqa: E402
from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv  # noqa: E402
from pfrl_amd.initializers import init_chainer_default  # noqa: E402
from pfrl_amd.q_functions import DiscreteActionValueHead  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--outdir", type=str, default="results")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--steps", type=int, default=10 ** 5)
    parser.add_argument("--num-envs", type=int, default=32)
    parser.add_argument("--replay-start-size", type=int, default=1000)
    parser.add_argument("--capacity", type=int, default=10 ** 5)
    parser.add_argument("--target-update-interval", type=int, default=2000)
    parser.add_argument("--update-interval", type=int, default=4)
    parser.add_argument("--batch-size", type=int, default=32)
    parser.add_argument("--episodic-update-len", type=int, default=10)
    parser.add_argument("--final-exploration-frames", type=int, default=10 ** 4)
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--p-done", type=float, default=1.0 / 500,
                        help="per-step termination probability of the synthetic episodes")
    args = parser.parse_args()
    logging.basicConfig(level=logging.INFO)
    utils.set_random_seed(args.seed)
    outdir = experiments.prepare_output_dir(args, args.outdir)

    n_actions = 6
    env = HostSyntheticAtariVectorEnv(args.num_envs, seed=args.seed, n_actions=n_actions,
                                      p_done=args.p_done)
    # train_drqn_ale.py:150-166: conv trunk, LSTM, linear head
    q_func = pfrl.nn.RecurrentSequential(
        nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(),
        nn.Conv2d(32, 64, 4, stride=2), nn.ReLU(),
        nn.Conv2d(64, 64, 3, stride=1), nn.Flatten(), nn.ReLU(),
        nn.LSTM(input_size=3136, hidden_size=512),
        init_chainer_default(nn.Linear(512, n_actions)), DiscreteActionValueHead())
    import torch

    opt = torch.optim.Adam(q_func.parameters(), lr=args.lr, eps=1e-4)
    rbuf = replay_buffers.EpisodicReplayBuffer(args.capacity)
    explorer = explorers.LinearDecayEpsilonGreedy(
        1.0, 0.01, args.final_exploration_frames, lambda: np.random.randint(n_actions))

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    agent = agents.DoubleDQN(
        q_func, opt, rbuf, gpu=args.gpu, gamma=0.99, explorer=explorer,
        replay_start_size=args.replay_start_size,
        target_update_interval=args.target_update_interval, update_interval=args.update_interval,
        minibatch_size=args.batch_size, batch_accumulator="mean", phi=phi, recurrent=True,
        episodic_update_len=args.episodic_update_len)
    experiments.train_agent_batch(agent, env, args.steps, outdir, log_interval=1000)
    print("statistics:", agent.get_statistics(), "episodes stored:", rbuf.n_episodes)


if __name__ == "__main__":
    main()
