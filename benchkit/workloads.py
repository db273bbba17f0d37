"""The workloads of BASELINE.json's configs, built as the reference's examples build them (bench.py)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchkit.supervisor import _tick

def build_rainbow(args, device, rank):
    """BASELINE configs[2]: examples/atari/reproduction/rainbow/train_rainbow.py:110-159 --
    CategoricalDoubleDQN, DistributionalDuelingDQN(51 atoms, [-10, 10]) with factorised
    NoisyNet (sigma 0.5), PrioritizedReplayBuffer(alpha 0.5, beta0 0.4, num_steps 3,
    normalize_by_max='memory'), Adam(6.25e-5, eps 1.5e-4), Greedy explorer."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.q_functions import DistributionalDuelingDQN

    N, n_actions = args.num_envs, 6
    pfrl.utils.set_random_seed(args.seed * 64 + rank)
    q_func = DistributionalDuelingDQN(n_actions, 51, -10, 10)
    pfrl.nn.to_factorized_noisy(q_func, sigma_scale=0.5)
    if args.cudnn_benchmark:
        torch.backends.cudnn.benchmark = True
    if args.channels_last:
        q_func = q_func.to(memory_format=torch.channels_last)
    if args.torch_optimizer:
        opt = torch.optim.Adam(q_func.parameters(), 6.25e-5, eps=1.5 * 10 ** -4, fused=True)
    else:
        from pfrl_amd.optimizers import FusedAdam     # torch.optim.Adam's step as one launch

        opt = FusedAdam(q_func.parameters(), 6.25e-5, eps=1.5 * 10 ** -4)
    store = DeviceFrameStore(getattr(args, "frame_slots", None) or args.capacity + N * 24 + 8192,
                             (84, 84), torch.uint8, device, stack=4)
    env = SyntheticAtariVectorEnv(N, store=store, seed=args.seed, env_id0=rank * N,
                                  n_actions=n_actions)
    rbuf = replay_buffers.PrioritizedReplayBuffer(
        args.capacity, alpha=0.5, beta0=0.4, betasteps=2 * 10 ** 6, num_steps=3,
        normalize_by_max="memory", slack=getattr(args, "slack", None),
        priority_pow=getattr(args, "priority_pow", "device"))

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    agent = agents.CategoricalDoubleDQN(
        q_func, opt, rbuf, gpu=device.index, gamma=0.99, explorer=explorers.Greedy(),
        minibatch_size=args.minibatch,
        replay_start_size=getattr(args, "replay_start", None) or 2 * 10 ** 4,
        target_update_interval=32000, update_interval=args.update_interval,
        batch_accumulator="mean", phi=phi)
    agent.grad_reducer.broadcast_parameters(agent.model)
    agent.sync_target_network()
    return agent, env, rbuf


def build_ppo(args, device, rank):
    """BASELINE configs[3]: examples/atari/train_ppo_ale.py:247-264 model, Adam(2.5e-4,
    eps 1e-5), update_interval = N*128, minibatch 32*N, 4 epochs, clip 0.1, grad clip 0.5."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.initializers import init_lecun_normal
    from pfrl_amd.policies import SoftmaxCategoricalHead

    N, n_actions = args.num_envs, 6
    pfrl.utils.set_random_seed(args.seed * 64 + rank)

    def lecun_init(layer, gain=1):
        init_lecun_normal(layer.weight, gain)
        torch.nn.init.zeros_(layer.bias)
        return layer

    nn = torch.nn
    model = nn.Sequential(
        lecun_init(nn.Conv2d(4, 32, 8, stride=4)), nn.ReLU(),
        lecun_init(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
        lecun_init(nn.Conv2d(64, 64, 3, stride=1)), nn.ReLU(), nn.Flatten(),
        lecun_init(nn.Linear(3136, 512)), nn.ReLU(),
        pfrl.nn.Branched(
            nn.Sequential(lecun_init(nn.Linear(512, n_actions), 1e-2), SoftmaxCategoricalHead()),
            lecun_init(nn.Linear(512, 1))))
    if args.cudnn_benchmark:
        torch.backends.cudnn.benchmark = True
    if args.channels_last:
        # Conv2d + ReLU pairs of the Sequential -> MIOpen conv + one fused bias/ReLU
        # launch (same parameters, same state_dict)
        model = pfrl.nn.fuse_conv_bias_relu(model).to(memory_format=torch.channels_last)
        if os.environ.get("PFRL_PPO_TRUNK", "1") == "1":
            # conv stack + hidden layer as the f32 MFMA trunk kernels (csrc/qnet.hip)
            pfrl.nn.fuse_sequential_trunk(model)
    opt = torch.optim.Adam(model.parameters(), lr=2.5e-4, eps=1e-5, fused=True)
    T = 128
    store = DeviceFrameStore((T + 8) * N + 8192, (84, 84), torch.uint8, device, stack=4)
    env = SyntheticAtariVectorEnv(N, store=store, seed=args.seed, env_id0=rank * N,
                                  n_actions=n_actions)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    agent = agents.PPO(model, opt, gpu=device.index, phi=phi, update_interval=N * T,
                       minibatch_size=32 * N, epochs=4, clip_eps=0.1, clip_eps_vf=None,
                       standardize_advantages=True, entropy_coef=1e-2, max_grad_norm=0.5,
                       reuse_next_values=bool(getattr(args, "ppo_reuse_next_values", 0)))
    agent.grad_reducer.broadcast_parameters(agent.model)
    return agent, env, None


def build_sac(args, device, rank):
    """BASELINE configs[4]: examples/mujoco/reproduction/soft_actor_critic/
    train_soft_actor_critic.py:172-243 -- 256-256 MLP policy (squashed Gaussian) and twin Q,
    Adam(3e-4), ReplayBuffer(10**6), B=256, update_interval=1, learned temperature;
    Humanoid-shaped synthetic env (obs f32[376], action f32[17])."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents, replay_buffers
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv
    from torch import distributions as D

    N, obs_size, action_size = args.num_envs, 376, 17

    def squashed_diagonal_gaussian_head(x):
        # tanh-squashed diagonal Gaussian, log-scale clamped to [-20, 2]
        mean, log_scale = torch.chunk(x, 2, dim=1)
        scale = torch.sqrt(torch.exp(torch.clamp(log_scale, -20.0, 2.0) * 2))
        return D.transformed_distribution.TransformedDistribution(
            D.Independent(D.Normal(loc=mean, scale=scale), 1),
            [D.transforms.TanhTransform(cache_size=1)])
    pfrl.utils.set_random_seed(args.seed * 64 + rank)
    nn = torch.nn
    policy = nn.Sequential(nn.Linear(obs_size, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                           nn.Linear(256, action_size * 2),
                           pfrl.nn.Lambda(squashed_diagonal_gaussian_head))
    for i in (0, 2, 4):
        nn.init.xavier_uniform_(policy[i].weight)
    if args.torch_optimizer:
        Adam = lambda ps: torch.optim.Adam(ps, lr=3e-4, fused=True)
    else:
        from pfrl_amd.optimizers import FusedAdam

        Adam = lambda ps: FusedAdam(ps, lr=3e-4)   # torch.optim.Adam's step as one launch
    popt = Adam(policy.parameters())

    def make_q():
        q = nn.Sequential(pfrl.nn.ConcatObsAndAction(), nn.Linear(obs_size + action_size, 256),
                          nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1))
        for i in (1, 3, 5):
            nn.init.xavier_uniform_(q[i].weight)
        return q, Adam(q.parameters())

    q1, q1opt = make_q()
    q2, q2opt = make_q()
    env = HostSyntheticVectorObsEnv(N, obs_dim=obs_size, act_dim=action_size,
                                    seed=args.seed * 64 + rank)
    rbuf = replay_buffers.ReplayBuffer(args.capacity)
    agent = agents.SoftActorCritic(
        policy, q1, q2, popt, q1opt, q2opt, rbuf, gamma=0.99, gpu=device.index,
        replay_start_size=10000, minibatch_size=args.minibatch, update_interval=1,
        burnin_action_func=lambda: np.random.uniform(-1, 1, size=action_size).astype(np.float32),
        entropy_target=-action_size, temperature_optimizer_lr=3e-4)
    from pfrl_amd import distributed

    distributed.broadcast_agent(agent)
    return agent, env, rbuf


def build_agent(args, device, rank):
    if args.algo == "rainbow":
        return build_rainbow(args, device, rank)
    if args.algo == "ppo":
        return build_ppo(args, device, rank)
    if args.algo == "sac":
        return build_sac(args, device, rank)
    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.initializers import init_chainer_default
    from pfrl_amd.q_functions import DiscreteActionValueHead

    N = args.num_envs
    n_actions = 6
    pfrl.utils.set_random_seed(args.seed * 64 + rank)
    # examples/atari/train_dqn_batch_ale.py:35-41 (arch "nature")
    q_func = torch.nn.Sequential(
        pfrl.nn.LargeAtariCNN(),
        init_chainer_default(torch.nn.Linear(512, n_actions)),
        DiscreteActionValueHead(),
    )
    # ... :199-206
    from pfrl_amd.optimizers import FusedRMSprop

    opt_cls = torch.optim.RMSprop if args.torch_optimizer else FusedRMSprop
    opt = opt_cls(q_func.parameters(), lr=2.5e-4, alpha=0.95, momentum=0.0, eps=1e-2,
                  centered=True)
    if args.cudnn_benchmark:
        torch.backends.cudnn.benchmark = True
    if args.channels_last:
        q_func = q_func.to(memory_format=torch.channels_last)
    if args.host_env:
        from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

        env = HostSyntheticAtariVectorEnv(N, seed=args.seed * 64 + rank, n_actions=n_actions,
                                          frame_pool=4096)
    else:
        frame_slots = getattr(args, "frame_slots", None) or args.capacity + N * 16 + 8192
        store = DeviceFrameStore(frame_slots, (84, 84), torch.uint8, device, stack=4)
        env = SyntheticAtariVectorEnv(N, store=store, seed=args.seed, env_id0=rank * N,
                                      n_actions=n_actions)
    rbuf = replay_buffers.ReplayBuffer(args.capacity, num_steps=1, slack=getattr(args, "slack", None))
    explorer = explorers.LinearDecayEpsilonGreedy(
        1.0, 0.01, 10 ** 6, lambda: np.random.randint(n_actions))

    def phi(x):  # :229-231
        return np.asarray(x, dtype=np.float32) / 255

    agent = agents.DQN(
        q_func, opt, rbuf, gpu=device.index, gamma=0.99, explorer=explorer,
        replay_start_size=getattr(args, "replay_start", None) or 5 * 10 ** 4,
        target_update_interval=3 * 10 ** 4, clip_delta=True,
        update_interval=args.update_interval, minibatch_size=args.minibatch,
        batch_accumulator="sum", phi=phi)
    if args.chunks is not None:
        agent.step_fused_chunks = tuple(float(x) for x in args.chunks.split(",") if x)
        agent._chunks_set_by_caller = True
    agent.grad_reducer.broadcast_parameters(agent.model)
    agent.sync_target_network()
    return agent, env, rbuf


def workload_description(args, N, rbuf):
    if args.algo == "dqn":
        return ("BASELINE.json configs[1]: DQN Nature-CNN, %d synthetic Atari-shaped envs/GPU "
                "(84x84x4 u8), ReplayBuffer(%d) on device prefilled to %d, B=%d, update_interval=%d "
                "(replay ratio %.1f sampled transitions per env-step), RMSprop centered, "
                "batch_accumulator=sum%s" % (N, args.capacity, len(rbuf), args.minibatch,
                                             args.update_interval,
                                             args.minibatch / args.update_interval,
                                             "; HOST env: frames ingested over PCIe (not the "
                                             "headline)" if args.host_env else ""))
    if args.algo == "rainbow":
        return ("BASELINE.json configs[2]: CategoricalDoubleDQN + DistributionalDuelingDQN(51 atoms) "
                "+ NoisyNet, %d synthetic Atari-shaped envs/GPU, PrioritizedReplayBuffer(%d, "
                "alpha=0.5, beta0=0.4, num_steps=3, normalize_by_max=memory) with sum/min trees in "
                "HBM prefilled to %d, B=%d, update_interval=%d, Adam"
                % (N, args.capacity, len(rbuf), args.minibatch, args.update_interval))
    if args.algo == "sac":
        return ("BASELINE.json configs[4]: SAC, %d MuJoCo-shaped synthetic envs/GPU (obs f32[376], "
                "action f32[17]; host env, observations ingested over PCIe), ReplayBuffer(%d) fp32 "
                "on device prefilled to %d, B=%d, update_interval=1 (one update per env-step), "
                "256-256 MLPs, Adam" % (N, args.capacity, len(rbuf), args.minibatch))
    return ("BASELINE.json configs[3]: PPO, %d synthetic Atari-shaped envs/GPU x 128-step rollouts, "
            "%s, "
            "update_interval=%d, minibatch=%d, 4 epochs, GAE + advantage standardisation kernels, "
            "Adam" % (N, "reuse_next_values=True (opt-in: V(next_state) from the next step's V(state), "
                      "equal to f32 rounding only; SURVEY 8(d)'s 0.854 MB/env-step variant)"
                      if getattr(args, "ppo_reuse_next_values", 0) else
                      "reuse_next_values=False = the reference's value pass, V over states AND "
                      "next_states (pfrl/agents/ppo.py:110-142); rows the two passes share (a "
                      "next_state that IS the next step's state) are evaluated once, bit-identical "
                      "to the brute-force second pass (tests/test_bench_path_parity.py), so the "
                      "bytes MOVED are SURVEY 8(d)'s 0.854 MB/env-step and that is what is priced",
                      N * 128, 32 * N))


def one_step(agent, env, obss, num_envs):
    actions = agent.batch_act(obss)
    obss, rs, dones, infos = env.step(actions)
    resets = np.zeros(num_envs, dtype=bool)
    agent.batch_observe(obss, rs, dones, resets)
    not_end = np.logical_not(dones)
    return env.reset(not_end)


def prefill(agent, env, obss, num_envs, target):
    """Fill the replay buffer through the normal act/observe path with updates
    disabled (the timed region then runs at full-buffer steady state)."""
    saved = agent.replay_updater.replay_start_size
    agent.replay_updater.replay_start_size = 1 << 62
    while len(agent.replay_buffer) < target:
        _tick("prefill")
        obss = one_step(agent, env, obss, num_envs)
    agent.replay_updater.replay_start_size = saved
    return obss
