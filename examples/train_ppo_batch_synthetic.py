#!/usr/bin/env python
"""PPO on synthetic Atari-shaped envs -- the structure of the reference's
examples/atari/train_ppo_ale.py (model :247-264, agent :281-296, LR / clip decay hooks
:298-318) with `pfrl` replaced by `pfrl_amd` and the ALE env factory by the on-device
synthetic VectorEnv (there is no ALE in this image)."""
import argparse
import os
import sys

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfrl_amd as pfrl  # noqa: E402
from pfrl_amd import experiments, utils  # noqa: E402
from pfrl_amd.agents import PPO  # noqa: E402
from pfrl_amd.initializers import init_lecun_normal  # noqa: E402
from pfrl_amd.policies import SoftmaxCategoricalHead  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--num-envs", type=int, default=512)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--outdir", type=str, default="results")
    parser.add_argument("--steps", type=int, default=10 ** 7)
    parser.add_argument("--eval-interval", type=int, default=10 ** 6)
    parser.add_argument("--eval-n-runs", type=int, default=10)
    parser.add_argument("--lr", type=float, default=2.5e-4)
    parser.add_argument("--update-interval", type=int, default=None,
                        help="default: 128 steps of every env")
    parser.add_argument("--batchsize", type=int, default=None, help="default: 32 per env")
    parser.add_argument("--epochs", type=int, default=4)
    parser.add_argument("--channels-last", action="store_true", default=True)
    args = parser.parse_args()

    import logging

    logging.basicConfig(level=logging.INFO)
    utils.set_random_seed(args.seed)
    os.makedirs(args.outdir, exist_ok=True)
    device = torch.device("cuda", args.gpu)
    N = args.num_envs
    update_interval = args.update_interval or 128 * N
    batchsize = args.batchsize or 32 * N

    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv

    def make_batch_env(test):
        store = DeviceFrameStore((update_interval // N + 8) * N + 8192, (84, 84), torch.uint8,
                                 device, stack=4)
        return SyntheticAtariVectorEnv(N, store=store, seed=args.seed + (10 ** 6 if test else 0))

    n_actions = 6

    def lecun_init(layer, gain=1):
        init_lecun_normal(layer.weight, gain)
        nn.init.zeros_(layer.bias)
        return layer

    model = nn.Sequential(
        lecun_init(nn.Conv2d(4, 32, 8, stride=4)), nn.ReLU(),
        lecun_init(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
        lecun_init(nn.Conv2d(64, 64, 3, stride=1)), nn.ReLU(), nn.Flatten(),
        lecun_init(nn.Linear(3136, 512)), nn.ReLU(),
        pfrl.nn.Branched(
            nn.Sequential(lecun_init(nn.Linear(512, n_actions), 1e-2), SoftmaxCategoricalHead()),
            lecun_init(nn.Linear(512, 1))))
    if args.channels_last:
        # MIOpen's NHWC kernels; minibatches are then gathered straight into that layout
        model = pfrl.nn.fuse_conv_bias_relu(model).to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, eps=1e-5)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    agent = PPO(model, opt, gpu=args.gpu, phi=phi, update_interval=update_interval,
                minibatch_size=batchsize, epochs=args.epochs, clip_eps=0.1, clip_eps_vf=None,
                standardize_advantages=True, entropy_coef=1e-2, max_grad_norm=0.5)

    def lr_setter(env, agent, value):
        for group in agent.optimizer.param_groups:
            group["lr"] = value

    def clip_eps_setter(env, agent, value):
        agent.clip_eps = max(value, 1e-8)

    hooks = [experiments.LinearInterpolationHook(args.steps, args.lr, 0, lr_setter),
             experiments.LinearInterpolationHook(args.steps, 0.1, 0, clip_eps_setter)]
    experiments.train_agent_batch_with_evaluation(
        agent=agent, env=make_batch_env(False), eval_env=make_batch_env(True), outdir=args.outdir,
        steps=args.steps, eval_n_steps=None, eval_n_episodes=args.eval_n_runs,
        eval_interval=args.eval_interval, log_interval=10 ** 5, save_best_so_far_agent=False,
        step_hooks=hooks)


if __name__ == "__main__":
    main()
