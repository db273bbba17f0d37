#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the batched DQN hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: 256
Atari-shaped envs step once (84x84 uint8 frames generated on the device into
the HBM frame ring) -> batch_act (gather u8->f32, Q forward, eps-greedy) ->
batch_observe (256 appends, 64 updates at the reference schedule: sample 32,
fused batch_experiences gather, Huber loss, backward, centered RMSprop) ->
env.reset(not_end).  BASELINE.json config[1]: DQN, Nature CNN,
ReplayBuffer(10**6) prefilled to capacity, B=32, update_interval=4,
batch_accumulator='sum', fp32 network.  Nothing is skipped inside the timed
region.  N GPUs = N ranks, the metric's 256 envs sharded 256 / N per rank (strong
scaling, the default; --scaling weak keeps 256 per rank), per-GPU-local replay,
per update one flat RCCL all-reduce of the small gradients + an all-gather of the
hidden layer's batch matrices (pfrl_amd/distributed.py).

Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--algo", choices=["dqn", "rainbow", "ppo", "sac"], default="dqn",
                   help="dqn = BASELINE configs[1] (headline); rainbow = configs[2]; ppo = configs[3]; "
                        "sac = configs[4]")
    p.add_argument("--host-env", action="store_true",
                   help="dqn only: frames are produced on the HOST (numpy) and ingested over PCIe "
                        "through the pinned staging ring -- the PCIe-inclusive rate of DESIGN.md, "
                        "never the headline value")
    p.add_argument("--steps", type=int, default=None,
                   help="timed batched env steps (default 200; 128 = one rollout + update for ppo)")
    p.add_argument("--warmup", type=int, default=None,
                   help="untimed steps (default 5; 128 = one full rollout + update for ppo)")
    p.add_argument("--num-envs", type=int, default=None, help="default 256 (512 for ppo)")
    p.add_argument("--capacity", type=int, default=10 ** 6)
    p.add_argument("--minibatch", type=int, default=32)
    p.add_argument("--update-interval", type=int, default=4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--replay-start", type=int, default=None,
                   help="replay_start_size (default: the example scripts' 5e4 dqn / 2e4 rainbow / 1e4 sac)")
    p.add_argument("--frame-slots", type=int, default=None,
                   help="slots of the HBM frame ring (default: capacity + room for the windows of all envs)")
    p.add_argument("--slack", type=int, default=None,
                   help="spare rows of the entry / transition rings (default 65536)")
    p.add_argument("--prefill", type=int, default=None,
                   help="transitions to prefill (default: capacity, i.e. full buffer)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--allow-lib-override", action="store_true",
                   help="accept PFRL_AMD_LIB (an A/B build of the native library); recorded on the line")
    p.add_argument("--no-data-path-only", action="store_true",
                   help="dqn: skip the second measurement with a zero-FLOP q_function")
    p.add_argument("--ppo-reuse-next-values", type=int, default=0, choices=[0, 1],
                   help="ppo: 0 (default, the package's default) = the reference's value pass over "
                        "states AND next_states (pfrl/agents/ppo.py:110-142; rows the two passes "
                        "share are evaluated once where that is bit-identical); 1 = the opt-in "
                        "shortcut without that guarantee (PPO(reuse_next_values=True))")
    p.add_argument("--no-also", action="store_true",
                   help="dqn: do not append the PPO configs[3] measurement ('also') to the line")
    p.add_argument("--scaling", choices=["weak", "strong"], default=None,
                   help="N > 1: weak = --num-envs envs on EVERY GPU; strong = --num-envs envs "
                        "sharded N/G per GPU as BASELINE.json's metric and SURVEY.md 8(e) describe "
                        "(default: strong)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=14.0,
                   help="end-to-end sample of the reference on this box's host cores")
    p.add_argument("--cpu-baseline-threads", type=int, default=16,
                   help="torch CPU threads for the baseline's network (capped by the host; at "
                        "B=32 the Nature CNN is slower with all 128 threads than with 16)")
    p.add_argument("--no-cudnn-benchmark", dest="cudnn_benchmark", action="store_false",
                   help="do not let MIOpen search conv algorithms")
    p.add_argument("--nchw", dest="channels_last", action="store_false",
                   help="keep the network in NCHW (default: channels_last)")
    p.add_argument("--blas", choices=["default", "rocblas", "tunable"], default=None,
                   help="GEMM back-end for the PyTorch side: torch default (hipBLASLt), rocBLAS, "
                        "or torch's TunableOp (times every rocBLAS / hipBLASLt solution once per "
                        "shape and keeps the fastest); default: tunable for dqn and sac "
                        "(measured +3 % and 2.5x), torch default for rainbow and ppo (no gain)")
    p.add_argument("--chunks", type=str, default=None,
                   help="dqn: env-range cut points of the step-fused path as fractions, e.g. "
                        "'0.125' (default) or '' for one range")
    p.add_argument("--priority-pow", choices=["device", "device_cr", "host_libm"], default="device",
                   help="rainbow: where (clip(err) + eps) ** alpha is evaluated.  device = one "
                        "launch, glibc's powf restated on the device (bit-exact priority trees); "
                        "device_cr = one launch, correctly rounded power (<= 1 ulp from NumPy's "
                        "powf in <1 %% of inputs; rounds 1-2); host_libm = this host's libm as "
                        "NumPy does (one D2H per update)")
    p.add_argument("--torch-optimizer", action="store_true",
                   help="stock torch.optim.RMSprop instead of the fused HIP step")
    args = p.parse_args()
    if args.scaling is None:
        args.scaling = "strong"
    if args.steps is None:
        args.steps = 128 if args.algo == "ppo" else 200
    if args.warmup is None:
        args.warmup = 128 if args.algo == "ppo" else 5
    if args.algo == "ppo":
        # large-batch convs: MIOpen's default choices are already good and the
        # exhaustive search costs minutes of GPU time for ~-8 % throughput here
        args.cudnn_benchmark = False
    if args.blas is None:
        args.blas = "tunable" if args.algo in ("dqn", "sac") else "default"
    if args.num_envs is None:
        args.num_envs = {"ppo": 512, "sac": 64}.get(args.algo, 256)
    if args.algo == "sac" and args.minibatch == 32:
        args.minibatch = 256
    return args


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


def cpu_baseline(args, seconds):
    """The same workload through the CPU oracle (oracle/pfrl_oracle.c = plain C
    restatement of the reference's data path) plus the same network in torch
    CPU, on this box's host cores, for a bounded sample.  kind = "port"."""
    N, B = args.num_envs, args.minibatch
    avail = torch.get_num_threads()
    cores = max(1, min(avail, args.cpu_baseline_threads))
    torch.set_num_threads(cores)
    try:
        return _cpu_baseline_run(args, seconds, N, B, cores)
    finally:
        torch.set_num_threads(avail)


def _cpu_baseline_run(args, seconds, N, B, cores):
    import oracle
    import pfrl_amd as pfrl
    from pfrl_amd.agents.dqn import compute_value_loss
    from pfrl_amd.initializers import init_chainer_default
    from pfrl_amd.q_functions import DiscreteActionValueHead
    from pfrl_amd.utils.random import sample_n_k

    rs = np.random.RandomState(0)
    F = 20000
    frames = rs.randint(0, 256, size=(F, 84 * 84)).astype(np.uint8)
    cap = 100000  # host memory bound, stated in the sample description
    t_state = rs.randint(0, F, size=(cap, 4)).astype(np.int32)
    t_next = rs.randint(0, F, size=(cap, 4)).astype(np.int32)
    rewards = rs.choice([-1.0, 0.0, 1.0], size=cap)
    terms = (rs.rand(cap) < 0.002).astype(np.uint8)
    actions = rs.randint(0, 6, size=cap)
    torch.manual_seed(0)
    q = torch.nn.Sequential(pfrl.nn.LargeAtariCNN(),
                            init_chainer_default(torch.nn.Linear(512, 6)),
                            DiscreteActionValueHead())
    tq = torch.nn.Sequential(pfrl.nn.LargeAtariCNN(), torch.nn.Linear(512, 6),
                             DiscreteActionValueHead())
    tq.load_state_dict(q.state_dict())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
    n_updates_per_step = N // args.update_interval
    t0 = time.perf_counter()
    updates = 0
    data_s = 0.0
    done = False
    while not done:
        d0 = time.perf_counter()
        refs = rs.randint(0, F, size=(N, 4)).astype(np.int32)
        x = oracle.batch_states_u8(frames, refs, 255.0).reshape(N, 4, 84, 84)
        data_s += time.perf_counter() - d0
        with torch.no_grad():
            q(torch.from_numpy(x)).greedy_actions.numpy()
        for _ in range(n_updates_per_step):
            d0 = time.perf_counter()
            idx = sample_n_k(cap, B)
            ents = [[int(i)] for i in idx]
            sc = oracle.batch_experiences_scalars(ents, rewards, terms, 0.99, 1)
            s = oracle.batch_states_u8(frames, t_state[idx], 255.0).reshape(B, 4, 84, 84)
            ns = oracle.batch_states_u8(frames, t_next[idx], 255.0).reshape(B, 4, 84, 84)
            data_s += time.perf_counter() - d0
            qout = q(torch.from_numpy(s))
            y = qout.evaluate_actions(torch.from_numpy(actions[idx]))
            with torch.no_grad():
                nq = tq(torch.from_numpy(ns)).max
                t = (torch.from_numpy(sc["reward"]) + torch.from_numpy(sc["discount"])
                     * (1.0 - torch.from_numpy(sc["is_state_terminal"])) * nq)
            loss = compute_value_loss(y, t, True, "sum")
            opt.zero_grad()
            loss.backward()
            opt.step()
            updates += 1
            # the sample is bounded by time, at update granularity: a batched step is
            # 64 updates (several seconds on the host), so fractions of a step count
            if time.perf_counter() - t0 >= seconds:
                done = True
                break
    el = time.perf_counter() - t0
    steps = updates / n_updates_per_step
    return {
        "value": round(steps * N / el, 2), "unit": "env-steps/s", "cores": cores, "kind": "port",
        "data_path_only_value": round(steps * N / max(data_s, 1e-9), 2),
        "sample": "%.2f batched steps of %d envs (%d updates of B=%d) in %.1f s; oracle C data "
                  "path (single thread) + torch-CPU Nature CNN (%d threads); replay capacity 1e5 on "
                  "the host" % (steps, N, updates, B, el, cores),
    }


PROFILE_BATCH_EXPERIENCES, PROFILE_BATCH_STATES_U8, PROFILE_GAE_SCAN, PROFILE_ADV_STATS = 0, 1, 2, 3   # pfrl_amd.ops constants
PROFILE_BATCH_STATES_U8_RAW = 4


def compute_roofline(algo, all_us, all_units, all_kinds):
    """``roofline`` object for the dominant HIP kernel of the path: the fused
    batch_experiences gather for the replay agents, the batch_states gather (value
    pass + minibatches) for PPO.  Inputs: per-launch durations (us), unit counts
    and kinds as returned by ``ops.profile_collect(kind=None)``."""
    k, fb = 4, 84 * 84
    if algo == "ppo":
        kind, kname, unit_name = PROFILE_BATCH_STATES_U8, "k_batch_states_u8", "frames"
        # per gathered frame: fb bytes read as u8, 4*fb written as f32
        per_unit = fb + 4 * fb
        if PROFILE_BATCH_STATES_U8_RAW in all_kinds:
            # round 5: the network reads u8 NHWC4 pixels (phi in the first convolution's operand
            # loader, agents/ppo.py _u8_pixels), so the gather writes one byte per frame byte: the
            # path's gather IS this kernel, priced at what it has to move (2 bytes per frame byte;
            # SURVEY 8d's 5 bytes assume the fp32 copy that no longer exists)
            kind, kname, per_unit = PROFILE_BATCH_STATES_U8_RAW, "k_batch_states_u8_raw", fb + fb
    elif algo == "sac":
        kind, kname, unit_name = PROFILE_BATCH_EXPERIENCES, "k_batch_experiences", "entries"
        # per sampled entry: state + next_state f32[376] read and written, action f32[17]
        # read and written, reward/terminal/discount
        per_unit = 2 * (2 * 376 * 4) + 2 * 17 * 4 + 2 * 12
    else:
        kind, kname, unit_name = PROFILE_BATCH_EXPERIENCES, "k_batch_experiences", "entries"
        # per sampled entry: state + next_state, each k frames read as u8, written as f32
        per_unit = 2 * k * (fb + 4 * fb)
    k_us = [u for u, kd in zip(all_us, all_kinds) if kd == kind]
    k_units = [n for n, kd in zip(all_units, all_kinds) if kd == kind]
    if not k_us:
        return None
    scan = {}
    for skind, sname, sbytes, swhat in (
            (PROFILE_GAE_SCAN, "k_gae_scan_lds", 8 + 4 + 4 + 1 + 1 + 4 + 4,
             "per (t, env): reward f64 + v + next_v f32 + nonterminal + cut u8 read, adv + v_teacher "
             "f32 written"),
            (PROFILE_ADV_STATS, "k_adv_partial", 4, "per advantage: one f32 read")):
        s_us = [u for u, kd in zip(all_us, all_kinds) if kd == skind]
        s_units = [n for n, kd in zip(all_units, all_kinds) if kd == skind]
        if s_us:
            gbs = sbytes * sum(s_units) / (sum(s_us) * 1e-6) / 1e9
            scan[sname] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(gbs / HBM_PEAK_GBS, 5), "launches_timed": len(s_us),
                           "avg_launch_us": round(sum(s_us) / len(s_us), 2),
                           "elements_per_launch": int(s_units[0]), "bytes_per_element": sbytes,
                           "what": swhat}
    # The kernel is launched in a few shapes (DQN: a small and a large env range per
    # step; PPO: acting, value pass and minibatch gathers).  The roofline object
    # describes the shape that moves the most bytes; the aggregate over every timed
    # launch of the kernel is reported next to it.
    classes = {}
    for u, n in zip(k_us, k_units):
        c = classes.setdefault(n, [0, 0.0])
        c[0] += 1
        c[1] += u
    main_units = max(classes, key=lambda n: n * classes[n][0])
    n_main, us_main = classes[main_units]
    bytes_main = per_unit * main_units
    achieved = bytes_main * n_main / (us_main * 1e-6) / 1e9
    tot_bytes = sum(per_unit * b for b in k_units)
    tot_s = sum(k_us) * 1e-6
    roofline = {
        "bound": "hbm", "kernel": kname,
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "bytes_per_launch": int(bytes_main),
        "%s_per_launch" % unit_name: int(main_units),
        "avg_launch_us": round(us_main / n_main, 2), "launches_timed": n_main,
        "share_of_kernel_bytes": round(bytes_main * n_main / tot_bytes, 4),
        "all_launches": {
            "achieved": round(tot_bytes / tot_s / 1e9, 1), "launches": len(k_us),
            "shapes": {str(n): {"launches": c[0], "avg_launch_us": round(c[1] / c[0], 2)}
                       for n, c in sorted(classes.items())}},
        "timing": "hipEvent pair attached to each dispatch (hipExtLaunchKernelGGL) on the "
                  "launch stream, inside the timed region",
    }
    if scan:
        # the north star's named scan / reduction kernels: one launch each per rollout, a few MB --
        # latency-bound (the launch, not the bytes), reported against the same HBM roofline
        roofline["scan_kernels"] = scan
    # HBM traffic cannot be sampled from inside the process: it is taken from the
    # committed rocprofv3 --pmc passes of this same command
    # (profiles/rNN_pmc_gather.json, tools/pmc_gather.py), per launch shape.
    prof_dir = os.path.join(ROOT, "profiles")
    names = sorted((n for n in os.listdir(prof_dir) if n.endswith(("_pmc_gather.json", "_pmc_ppo.json", "_pmc_rainbow.json", "_pmc_sac.json"))),
                   reverse=True)      # newest round first
    # A PMC pass describes the build it was taken on: it carries the hash of the gather kernels'
    # sources (tools/pmc_gather.py: "kernel_sources_sha16") and is attached only while those files
    # are unchanged; otherwise traffic stays null and the line says which pass went stale.
    current = gather_sources_sha16()
    for name in names:
        try:
            pmc = json.load(open(os.path.join(prof_dir, name)))
            kk = pmc["kernels"].get("%s (%d %s)" % (kname, main_units, unit_name))
            if kk and "traffic_bytes_per_launch" in kk:
                taken_on = pmc.get("kernel_sources_sha16")
                if taken_on != current:
                    roofline["traffic_source"] = (
                        "none: profiles/%s was taken on gather sources %s, this build is %s"
                        % (name, taken_on or "of an untagged earlier round", current))
                    break
                roofline["traffic"] = kk["traffic_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / " \
                                             "WRITE_SIZE, separate passes, corrected; taken on " \
                                             "gather sources %s = this build)" % (name, taken_on)
                break
        except Exception:
            pass
    return roofline


def gather_sources_sha16():
    """sha256 (first 16 hex digits) over the sources of the gather kernels the roofline object
    describes: what a PMC pass is valid for."""
    import hashlib

    h = hashlib.sha256()
    for rel in ("pfrl_amd/csrc/replay.hip", "pfrl_amd/csrc/nhwc.h", "pfrl_amd/csrc/common.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


NATURE_FWD_FLOPS = 2 * (20 * 20 * 32 * 8 * 8 * 4 + 9 * 9 * 64 * 4 * 4 * 32 + 7 * 7 * 64 * 3 * 3 * 64
                        + 3136 * 512 + 512 * 6)      # per observation, pfrl/nn/atari_cnn.py:17-47
NATURE_CONV1_FLOPS = 2 * 20 * 20 * 32 * 8 * 8 * 4
MFMA_F32_PEAK_TFLOPS = 155.0    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, measured


def step_flops_dqn(N, minibatch, update_interval):
    """Arithmetic of one batched DQN step: acting forward on N observations + per update the
    online forward / backward on B (backward = 2 x forward minus conv1's input gradient, which
    does not exist) and the target forward on B."""
    updates = N // update_interval
    per_update = minibatch * (NATURE_FWD_FLOPS + 2 * NATURE_FWD_FLOPS - NATURE_CONV1_FLOPS
                              + NATURE_FWD_FLOPS)
    return N * NATURE_FWD_FLOPS + updates * per_update


def step_flops_ppo(N, n_actions=6, value_passes=2.0):
    """Arithmetic EXECUTED by one batched PPO step (512 envs), the rollout's passes amortised per
    env step: acting forward, the value pass(es) over the rollout (``value_passes``: 2 = states and
    all next_states; 1 + the fraction of next_states actually evaluated when shared rows are taken
    from the first pass), and 4 epochs of forward + backward (backward = 2 x forward minus conv1's
    input gradient); the two narrow heads (512 -> A, 512 -> 1) counted with the trunk."""
    heads = 2 * 512 * (n_actions + 1)
    fwd = NATURE_FWD_FLOPS + heads
    return N * (fwd + value_passes * fwd + 4 * (3 * fwd - NATURE_CONV1_FLOPS))


def mfma_per_launch(agent, rbuf, B=32, n_actions=6):
    """Per launch of ONE DQN update, MEASURED in this run: the launches a captured update replays
    (``GraphedUpdate.measure_launches``: the same Python run eagerly on a fresh minibatch, every
    library entry point bracketed by a pair of timing events, median of 5) and the arithmetic each
    performs (the Nature CNN of pfrl/nn/atari_cnn.py:17-47 at minibatch B: forward per layer; a
    backward launch = input gradient + weight gradient of its layer = 2 x forward, conv1 has no
    input gradient; the hidden layer's RMSprop step rides in the last backward launch), as a
    fraction of the f32 MFMA peak.  Eager durations carry ~1 us of event bracketing each and no
    graph-internal boundary: their sum is not ``update_us`` (that is the range graph's own clock)."""
    conv1 = 2 * 20 * 20 * 32 * 8 * 8 * 4
    conv2 = 2 * 9 * 9 * 64 * 4 * 4 * 32
    conv3 = 2 * 7 * 7 * 64 * 3 * 3 * 64
    hidden = 2 * 3136 * 512
    head = 2 * 512 * n_actions
    fwd = iter([(conv1, "conv1 fwd"), (conv2, "conv2 fwd"), (conv3, "conv3 fwd"), (hidden, "hidden fwd")])
    bwd = iter([(2 * hidden, "hidden bwd (input + weight gradient)"), (2 * conv3, "conv3 bwd"),
                (2 * conv2, "conv2 bwd")])
    seqs = [rbuf.lookahead_sample(B)]
    big = rbuf.fetch_many(seqs, agent.phi, agent.gamma)
    ns = big["next_state"]
    raw = agent._precompute_target_raw(ns.view((B,) + tuple(ns.shape[2:])))
    big["target_next_raw"] = raw.view((1, B) + tuple(raw.shape[1:]))
    calls = agent._graphed.measure_launches({k: v[0] for k, v in big.items()})
    out = []
    for name, us in calls:
        if name == "pfrl_conv2d_nhwc_fwd":
            f, w = next(fwd, (0, "forward"))
        elif name == "pfrl_dqn_head_td_loss":
            f, w = 3 * head, "hidden-layer fold + head + TD loss + head bwd"
        elif name == "pfrl_conv2d_nhwc_bwd":
            f, w = next(bwd, (0, "backward"))
        elif name == "pfrl_conv2d_nhwc_bwd_weight_ride":
            f, w = conv1, "conv1 wgrad (+ the hidden layer's RMSprop step riding)"
        elif name == "pfrl_rmsprop_fused_step":
            f, w = 0, "RMSprop (slab folds + step of the convolutions and the head)"
        else:
            f, w = 0, "-"
        gf = f * B / 1e9
        out.append({"entry": name, "what": w, "us": round(us, 2), "gflop": round(gf, 4),
                    "frac": round(gf / 1e3 / (us * 1e-6) / MFMA_F32_PEAK_TFLOPS, 4) if us > 0 and f else None})
    return {"source": "measured in this run (hipEvent pair around each launch of one eager update)",
            "launches": out, "n_launches": len(out),
            "sum_us": round(sum(o["us"] for o in out), 1)}


def launches_per_update():
    """Kernel launches of one update, from the committed rocprofv3 timeline of this build
    (profiles/rNN_dqn_update_timeline.txt, tools/update_timeline.py), newest round first."""
    import re

    prof_dir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof_dir), reverse=True):
        if name.endswith("_dqn_update_timeline.txt"):
            m = re.search(r"kernels (\d+),", open(os.path.join(prof_dir, name)).read())
            if m:
                return {"value": int(m.group(1)), "source": "profiles/" + name}
    return None


def algorithmic_bytes_per_step(algo, N, minibatch, update_interval, value_passes=2.0):
    """SURVEY.md 8(d) per env-step figures x envs per batched step."""
    fb, k = 84 * 84, 4
    if algo in ("dqn", "rainbow"):
        rho = minibatch / update_interval
        return N * (fb + (k * fb + 4 * k * fb) + rho * 2 * (k * fb + 4 * k * fb))
    if algo == "ppo":
        # act 141,120 + ring 7,056 + value pass + 4 epochs x 141,120 + GAE 24 +
        # adv-norm 12.  The reference evaluates V on states AND next_states (2 x 141,120: 994,932 B);
        # priced here are the bytes the build MOVES: next_states that are the next step's state are
        # not gathered again (value_passes = 1 + evaluated fraction: the 0.854 MB variant SURVEY
        # says to flag -- flagged in config.workload; the VALUES are the full second pass's).
        return N * (141120 + 7056 + value_passes * 141120 + 4 * 141120 + 24 + 12)
    return N * (2 * minibatch * 3084 + 3084)   # sac


def assemble_result(args, world, N, elapsed, n_updates, t_fill, workload, roofline):
    """The ONE JSON line of the driver contract (cpu_baseline etc. are added by the caller)."""
    ms = elapsed / args.steps * 1e3
    if roofline is not None:
        # the step as a whole against the same roofline: the kernel fraction above is for
        # the dominant gather alone and must not be read as the end-to-end figure
        nvp = getattr(args, "_ppo_next_value_pass", None) or {}
        vp = 1.0 + (nvp.get("evaluated", nvp.get("of", 1)) / max(1, nvp.get("of", 1)))
        step_bytes = algorithmic_bytes_per_step(args.algo, N, args.minibatch, args.update_interval,
                                                vp)
        roofline["step_algorithmic_bytes"] = int(step_bytes)
        roofline["step_frac"] = round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if args.algo == "dqn":
            # the step against the OTHER roofline: its arithmetic / the f32 MFMA peak (the data
            # path is HBM-bound, the step as a whole is bound by the B = 32 update chain)
            fl = step_flops_dqn(N, args.minibatch, args.update_interval)
            tf = fl / (ms * 1e-3) / 1e12
            roofline["mfma"] = {"step_flops": int(fl), "achieved": round(tf, 2),
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
            # what fp32 arithmetic + the reference's schedule allow at all: the N / update_interval
            # updates of a batched step are DEPENDENT (each reads the parameters the previous one
            # wrote, pfrl/agents/dqn.py:516-549, pfrl/replay_buffer.py:329-356), so even with
            # every launch at the f32 MFMA peak a step takes step_flops / peak
            per_update = (fl - N * NATURE_FWD_FLOPS) / max(1, N // args.update_interval)
            floor_us = per_update / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e6
            ceiling = N / (fl / (MFMA_F32_PEAK_TFLOPS * 1e12))
            roofline["mfma"].update({
                "flop_per_update": int(per_update), "floor_us_per_update": round(floor_us, 2),
                "parity_ceiling_env_steps_s": int(ceiling),
                "frac_of_parity_ceiling": round(N / (ms * 1e-3) / ceiling, 4),
                "ceiling_what": "env-steps/s if every launch of the step ran at the f32 MFMA peak under "
                                "the reference's schedule (B = %d, %d dependent updates per %d-env step, "
                                "fp32 as the parity contract demands); the north star's 1 M env-steps/s "
                                "is above it" % (args.minibatch, N // args.update_interval, N)})
        if args.algo == "ppo":
            # PPO is bound by the f32 MFMA trunk at update size (B = 16384), not by the gather the
            # HBM block above describes (3 % of the device time): rollout FLOPs / time / peak
            fl = step_flops_ppo(N, value_passes=vp)
            tf = fl / (ms * 1e-3) / 1e12
            roofline["mfma"] = {"step_flops": int(fl), "achieved": round(tf, 2),
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                "value_passes": round(vp, 4), "next_value_pass": nvp or None,
                                "what": "acting + value pass(es) + 4 epochs x (forward + backward) of the "
                                        "rollout, per env step (FLOPs executed); per-layer fractions at B = 16384: "
                                        "profiles/r04_layer_final.txt (tools/layer_bench.py)"}
            roofline["captured_launches_not_timed"] = (
                "the 65536-frame minibatch gathers run inside the captured update graph "
                "(agents/ppo.py::_minibatch_step) and carry no event pair; the timed launches are the "
                "value pass's gathers of the same kernel")
    per = "per GPU" if args.scaling == "weak" else "sharded over the GPUs"
    return {
        "metric": "env-steps/sec whole node (%s %d envs %s)" % (args.algo.upper(), N if args.scaling == "weak" else N * world, per),
        "value": round(world * N * args.steps / elapsed, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (host frames over PCIe)" if args.host_env else "synthetic",
        "config": {
            "workload": workload,
            "global_envs": world * N, "envs_per_gpu": N, "updates_in_timed_region": n_updates,
            "parallelism": "env-sharded dp%d, per-GPU-local replay" % world,
            "prefill_s": round(t_fill, 1),
        },
        "roofline": roofline,
    }


class _ZeroFlopQ(torch.nn.Module):
    """q_function stand-in for the data-path-only figure: Q-values that do
    not depend on the observation, one learnable row, so that every replay / gather / loss /
    optimizer launch of the step still happens and the network costs nothing."""

    def __init__(self, n_actions):
        super().__init__()
        self.q = torch.nn.Parameter(torch.zeros(1, n_actions))

    def forward(self, x):
        from pfrl_amd.action_value import DiscreteActionValue

        return DiscreteActionValue(self.q.expand(x.shape[0], self.q.shape[1]))


def data_path_only(args, device, agent, env, rbuf, obss, steps):
    """The same batched step over the same (full) replay buffer and env with a zero-FLOP
    q_function: appends, index draws, the fused gathers, TD loss and optimizer step remain."""
    from pfrl_amd import agents
    from pfrl_amd.optimizers import FusedRMSprop

    N = args.num_envs
    q = _ZeroFlopQ(6)
    opt = FusedRMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
    stub = agents.DQN(q, opt, rbuf, gpu=device.index, gamma=0.99, explorer=agent.explorer,
                      replay_start_size=agent.replay_start_size,
                      target_update_interval=3 * 10 ** 4, clip_delta=True,
                      update_interval=args.update_interval, minibatch_size=args.minibatch,
                      batch_accumulator="sum", phi=agent.phi)
    stub.step_fused_chunks = ()   # nothing to overlap host preparation with: one range
    stub.t = agent.t
    for _ in range(3):
        obss = one_step(stub, env, obss, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        obss = one_step(stub, env, obss, N)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"value": round(N * steps / el, 1), "unit": "env-steps/s", "steps": steps,
           "ms_per_step": round(el / steps * 1e3, 3),
           "what": "the same step with a zero-FLOP q_function (SURVEY.md 8d): env frames, "
                   "act gather, appends, index draws, fused minibatch gathers of the full "
                   "schedule, TD loss, optimizer step on one row"}
    # ... and with the per-update launches gone too (the zero-FLOP network still costs 4 launches
    # per update, 256 per step, which is all that bounds the figure above): what the replay side
    # ALONE sustains -- env frames, the acting gather + action select, the native planner, one
    # transfer, appends and the 2 048-entry gather of the step's whole schedule.

    class _NoUpdates:
        graphs = {("range",): None}
        pipeline = False

        def range_capturable(self):
            return True

        zeros = {}

        def run_range(self, big):
            U, B = big["reward"].shape[:2]
            z = self.zeros.get((U, B))
            if z is None:
                z = self.zeros[(U, B)] = (torch.zeros(U, device=device),
                                          torch.zeros(U * B, device=device))
            return z

    single = not (torch.distributed.is_available() and torch.distributed.is_initialized()
                  and torch.distributed.get_world_size() > 1)
    if not (single and stub.use_graphs and stub.range_graphs):
        return out, obss       # (the range-graph path is what the stand-in below replaces)
    try:
        stub._graphed = _NoUpdates()
        stub.batch_target_pass = False
        stub.target_update_interval = 10 ** 12   # (no sync inside a range: every range is "one graph")
        for _ in range(3):
            obss = one_step(stub, env, obss, N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            obss = one_step(stub, env, obss, N)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        out["without_update_launches"] = {
            "value": round(N * steps / el2, 1), "ms_per_step": round(el2 / steps * 1e3, 3),
            "what": "the replay side alone: env frames, acting gather + action select, native "
                    "planner + one transfer, appends, the fused gather of all 64 minibatches of the "
                    "step; no per-update launch"}
        data_path_only.last_stub = stub          # (tools/data_path_phases.py times its phases)
    except Exception as e:      # an extra figure must never cost the line its numbers
        sys.stderr.write("data_path_only.without_update_launches failed: %r\n" % (e,))
    return out, obss


def reference_baseline(args):
    """pfnet/pfrl ITSELF (gpu=-1) on the same synthetic workload, timed on THIS box's host cores
    by tools/reference_cpu_baseline.py in a subprocess that imports the reference from
    oracle/_ref/ (its modules compiled to .pyc by oracle/build_ref.py; /root/reference does not
    exist on the GPU box).  A bounded sample: BASELINE.md section 3's full protocol (>= 2e4
    env-steps, 3 seeds) is profiles/r03_reference_cpu_baseline_gpubox.json."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref_dir, "pfrl")):
        return None
    env = dict(os.environ, PFRL_REFERENCE=ref_dir, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "reference_cpu_baseline.py"),
           "--seconds", str(args.cpu_baseline_seconds), "--dp-seconds",
           str(max(2.0, args.cpu_baseline_seconds * 0.4)), "--prefill", "5120",
           "--threads", str(args.cpu_baseline_threads), "--num-envs", str(args.num_envs)]
    try:
        out = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:     # the baseline must never cost the line its GPU numbers
        sys.stderr.write("reference cpu baseline failed: %r\n" % (e,))
        return None
    return {
        "value": d["end_to_end"]["value"], "unit": "env-steps/s", "cores": d["cores"],
        "kind": "reference",
        "data_path_only_value": d["data_path_only"]["value"],
        "host_cores": d["host_cores"],
        "sample": "pfnet/pfrl itself (compiled from /root/reference into oracle/_ref), gpu=-1, "
                  "train loop of pfrl/agents/dqn.py on %d in-process synthetic Atari-shaped envs, "
                  "ReplayBuffer(1e5) holding %d transitions at the start: %d env-steps end to end "
                  "in %.0f s with %d torch threads, %d env-steps with a zero-FLOP q_function; "
                  "full protocol (>= 2e4 env-steps, 3 seeds, median): profiles/"
                  "r03_reference_cpu_baseline_gpubox.json"
                  % (d["num_envs"], d["replay_len_at_start"],
                     d["end_to_end"]["env_steps_per_sample"][0], args.cpu_baseline_seconds,
                     d["cores"], d["data_path_only"]["env_steps_per_sample"][0]),
    }


def reference_baseline_ppo(args, num_envs=512, steps=16):
    """The reference's PPO (pfrl/agents/ppo.py:465-532, gpu=-1, the model and hyperparameters of
    examples/atari/train_ppo_ale.py:247-264) on this box's host cores, by the same tool and the same
    oracle/_ref/ copy as :func:`reference_baseline`.  Bounded: rollouts of ``steps`` steps instead of
    128 (update_interval and minibatch scaled with them: every transition still gets one acting
    forward, one value pass and 4 epochs), one untimed + one timed rollout INCLUDING its update.
    The full-size figure (128-step rollouts) is profiles/r04_reference_cpu_baseline_ppo_gpubox.json."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref_dir, "pfrl")):
        return None
    env = dict(os.environ, PFRL_REFERENCE=ref_dir, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "reference_cpu_baseline.py"), "--algo", "ppo",
           "--num-envs", str(num_envs), "--ppo-steps", str(steps), "--ppo-rollouts", "1",
           "--threads", str(args.cpu_baseline_threads)]
    try:
        out = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:     # the baseline must never cost the line its GPU numbers
        sys.stderr.write("reference PPO cpu baseline failed: %r\n" % (e,))
        return None
    return {
        "value": d["end_to_end"]["value"], "unit": "env-steps/s", "cores": d["cores"],
        "kind": "reference", "host_cores": d["host_cores"],
        "sample": "pfnet/pfrl itself (oracle/_ref), gpu=-1, pfrl/agents/ppo.py on %d in-process "
                  "synthetic Atari-shaped envs: one %d-step rollout INCLUDING its update "
                  "(update_interval=%d, minibatch=%d, 4 epochs; BASELINE's rollout is 128 steps, the "
                  "per-transition work is the same): %d env-steps in %.0f s with %d torch threads; "
                  "full size: profiles/r04_reference_cpu_baseline_ppo_gpubox.json"
                  % (d["num_envs"], d["rollout_steps"], d["update_interval"], d["minibatch"],
                     d["end_to_end"]["env_steps"], d["end_to_end"]["seconds"], d["cores"]),
    }


def reference_baseline_other(args, algo, num_envs, seconds=8.0):
    """The reference's Rainbow / SAC (gpu=-1; tools/reference_cpu_baseline.py --algo rainbow|sac, the
    constructions of train_rainbow.py:110-159 / train_soft_actor_critic.py:172-243) on this box's host
    cores from the same oracle/_ref/ copy: a bounded sample of whole env steps with their updates."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref_dir, "pfrl")):
        return None
    env = dict(os.environ, PFRL_REFERENCE=ref_dir, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "reference_cpu_baseline.py"), "--algo", algo,
           "--num-envs", str(num_envs), "--seconds", str(seconds), "--prefill", "5120",
           "--threads", str(args.cpu_baseline_threads)]
    try:
        out = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:     # the baseline must never cost the line its GPU numbers
        sys.stderr.write("reference %s cpu baseline failed: %r\n" % (algo, e))
        return None
    return {
        "value": d["end_to_end"]["value"], "unit": "env-steps/s", "cores": d["cores"],
        "kind": "reference", "host_cores": d["host_cores"],
        "sample": "pfnet/pfrl itself (oracle/_ref), gpu=-1, %s on %d in-process synthetic envs, replay "
                  "capacity 1e5 holding %d transitions at the start: %d env-steps (%d updates) in %.0f s "
                  "with %d torch threads"
                  % (d["what"].split("pfrl ")[1].split(" (")[0], d["num_envs"], d["replay_len_at_start"],
                     d["end_to_end"]["env_steps"], d["end_to_end"]["updates"],
                     d["end_to_end"]["seconds"], d["cores"]),
    }


def collective_microbench(agent, device, iters=50):
    """Per-update exchange of the data-parallel update, timed on its own (hipEvents around ``iters``
    back-to-back calls, every rank takes part): the flat bucket's all-reduce and the grouped
    all-gather of the large Linear layer's batch matrices at this agent's sizes.  The in-graph cost
    is part of ``update_us``; these are the collectives' stand-alone latencies."""
    red = getattr(agent, "grad_reducer", None)
    comm = getattr(red, "_comm", None)
    if red is None or not red.active():
        return None
    out = {}
    bucket = red.current_bucket() if red.current_bucket() is not None else red._flat
    try:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if bucket is not None:
            buf = torch.zeros_like(bucket)
            for _ in range(3):
                red.reduce_flat(buf)
            ev[0].record()
            for _ in range(iters):
                red.reduce_flat(buf)
            ev[1].record()
            torch.cuda.synchronize()
            out["flat_all_reduce"] = round(ev[0].elapsed_time(ev[1]) * 1e3 / iters, 2)
            out["flat_bytes"] = int(buf.numel() * 4)
        from pfrl_amd import distributed

        B = getattr(agent, "minibatch_size", 32)
        if comm is not None and distributed.lowrank_pays(B, 512, 3136, comm.world):
            G = comm.world
            dy, x = torch.zeros(B * 512, device=device), torch.zeros(B * 3136, device=device)
            dya, xa = torch.zeros(G * B * 512, device=device), torch.zeros(G * B * 3136, device=device)
            for i in range(iters + 3):
                if i == 3:
                    ev[2].record()
                with comm.group():
                    comm.all_gather(dya, dy)
                    comm.all_gather(xa, x)
            ev[3].record()
            torch.cuda.synchronize()
            out["lowrank_all_gather_pair"] = round(ev[2].elapsed_time(ev[3]) * 1e3 / iters, 2)
            out["lowrank_bytes_per_rank"] = int((dy.numel() + x.numel()) * 4)
    except Exception as e:      # (evidence only: never costs the line)
        out["note"] = "not measured: %s" % (str(e)[:120],)
    return out


class _StallWatchdog:
    """N > 1 only: if the ranks stop making progress (a peer died, a collective hangs), rank 0
    still prints a line -- ``value`` null, ``config.dp_plan`` = "fallback:stalled ..." -- and the
    process leaves with status 0 instead of sitting in the driver's timeout."""

    def __init__(self, args, rank, world, result_fd, limit_s):
        import threading

        self.t = time.time()
        self.what = "start"
        self.done = False
        self.printed = False
        self.args, self.rank, self.world, self.fd, self.limit = args, rank, world, result_fd, limit_s
        threading.Thread(target=self._run, name="pfrl-bench-watchdog", daemon=True).start()

    def tick(self, what):
        self.t, self.what = time.time(), what

    def _run(self):
        while not self.done:
            time.sleep(1.0)
            if time.time() - self.t > self.limit:
                a = self.args
                line = {"metric": "env-steps/sec whole node (%s)" % a.algo.upper(), "value": None,
                        "unit": "env-steps/s", "n_gpus": self.world, "steps": a.steps,
                        "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                        "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                        "config": {"workload": "not completed", "ranks_seen": self.world,
                                   "dp_plan": "fallback:stalled for %.0f s in %s" % (self.limit, self.what)}}
                if self.rank == 0 and not self.printed:
                    os.write(self.fd, (json.dumps(line) + "\n").encode())
                os._exit(0)


_WATCHDOG = [None]


def _tick(what):
    if _WATCHDOG[0] is not None:
        _WATCHDOG[0].tick(what)


def run_workload(args, device, rank, world, result_extras=True):
    """Build, prefill, warm up and time one workload; returns the result dict (rank 0) or None."""
    from pfrl_amd import ops

    _tick("build %s" % args.algo)
    agent, env, rbuf = build_agent(args, device, rank)
    N = args.num_envs
    obss = env.reset()
    _tick("prefill %s" % args.algo)
    t_fill = time.perf_counter()
    if rbuf is not None:
        # every replay workload runs at its stated size: the buffer is FULL when the timed region
        # starts (configs[4]: 1M fp32 transitions = 3.1 GB of rows, far outside the 256 MiB
        # Infinity Cache; round 3 timed SAC on a 10 % fill)
        target = args.prefill if args.prefill is not None else args.capacity
        target = max(min(target, args.capacity),
                     getattr(agent, "replay_start_size", None)
                     or agent.replay_updater.replay_start_size)
        obss = prefill(agent, env, obss, N, target)
    torch.cuda.synchronize()
    t_fill = time.perf_counter() - t_fill

    def barrier():
        # (the process group is the control plane -- gloo by default, pfrl_amd/distributed.py -- so the
        # barrier is a host rendezvous: the device is drained before AND after it)
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    # One-time work of the first updating steps (HIP-graph capture of every minibatch
    # buffer, MIOpen solver search, TunableOp GEMM tuning) must not fall into the timed
    # region even if the caller asks for fewer than 3 warm-up steps; for PPO the same
    # holds for the first rollout + update.
    need = (128 if args.algo == "ppo" else 3) - args.warmup
    for _ in range(max(0, need)):
        _tick("warm-up %s" % args.algo)
        obss = one_step(agent, env, obss, N)
    for _ in range(args.warmup):
        _tick("warm-up %s" % args.algo)
        obss = one_step(agent, env, obss, N)

    def updates_done():
        for name in ("optim_t", "n_updates", "n_policy_updates"):
            if hasattr(agent, name):
                return getattr(agent, name)
        return 0

    optim_before = updates_done()
    ops.profile_collect(kind=None)      # drop launches timed by an earlier workload
    ops.profile_enable(True)
    graphed = getattr(agent, "_graphed", None)
    if graphed is not None and hasattr(graphed, "run_range"):
        graphed.time_ranges = []
    barrier()
    torch.cuda.synchronize()
    # (one event per step boundary: where the device time of the timed region goes, step by step)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(min(args.steps, 1024) + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        _tick("timed steps %s" % args.algo)
        obss = one_step(agent, env, obss, N)
        if i + 1 < len(marks):
            marks[i + 1].record()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    _tick("after timed steps %s" % args.algo)
    ops.profile_enable(False)
    n_updates = updates_done() - optim_before

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        if torch.distributed.get_backend() == "nccl":
            tmax = tmax.to(device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    all_us, all_units, all_kinds = ops.profile_collect(kind=None)
    roofline = compute_roofline(args.algo, all_us, all_units, all_kinds)
    args._ppo_next_value_pass = getattr(agent, "next_value_pass", None)
    out = assemble_result(args, world, N, elapsed, n_updates, t_fill,
                          workload_description(args, N, rbuf), roofline)
    if graphed is not None and getattr(graphed, "time_ranges", None):
        timed, graphed.time_ranges = graphed.time_ranges, None
        big_u = max(u for u, _, _ in timed)
        us = [e0.elapsed_time(e1) * 1e3 / u for u, e0, e1 in timed if u == big_u]
        if roofline is not None and us:
            roofline.setdefault("mfma", {})["update_us"] = round(sum(us) / len(us), 2)
            roofline["mfma"]["update_us_what"] = (
                "device time of ONE optimizer update (forward, TD loss, backward, optimizer step) "
                "inside the captured %d-update range graph, hipEvents around the replay" % big_u)
            try:
                if args.algo != "dqn" or torch.distributed.is_initialized():
                    raise RuntimeError("measured for the single-process DQN update only")
                pl = mfma_per_launch(agent, rbuf, args.minibatch)
                roofline["mfma"]["per_launch"] = pl
                roofline["mfma"]["launches_per_update"] = {"value": pl["n_launches"], "source": pl["source"]}
            except Exception as e:      # (the measurement must not cost the line)
                roofline["mfma"]["per_launch"] = {"note": "not available: %s" % e}
                roofline["mfma"]["launches_per_update"] = launches_per_update()
    out["config"]["ranks_seen"] = world
    if step_ms:
        srt = sorted(step_ms)
        split = {"median_step_ms": round(srt[len(srt) // 2], 4), "max_step_ms": round(srt[-1], 4),
                 "what": "device time between step boundaries (hipEvents), timed region"}
        if args.algo == "ppo" and len(srt) >= 8:
            # a rollout = (steps - 1) acting steps + the step that also runs the update
            n_up = max(1, args.steps * N // (N * 128))
            split["act_phase_ms"] = round(sum(srt[:-n_up]), 3)
            split["update_phase_ms"] = round(sum(srt[-n_up:]) - n_up * srt[len(srt) // 2], 3)
        out["config"]["phase_split"] = split
    if torch.distributed.is_initialized():
        from pfrl_amd import rccl

        out["config"].update(rccl.status())
        out["config"]["collective_us"] = collective_microbench(agent, device)
    if args.algo == "rainbow":
        out["config"]["priority_pow"] = rbuf.priority_pow
    if result_extras and args.algo == "dqn" and not args.host_env and not args.no_data_path_only:
        # every rank runs it (the agents are symmetric); rank 0 reports its own
        dpo, obss = data_path_only(args, device, agent, env, rbuf, obss, min(args.steps, 100))
        out["data_path_only"] = dpo
    del agent, env, rbuf
    torch.cuda.empty_cache()
    return out if rank == 0 else None


def _die_with_parent():
    import ctypes
    import signal

    ctypes.CDLL("libc.so.6").prctl(1, signal.SIGKILL)       # PR_SET_PDEATHSIG


# what the ranks try, in order, when a data-parallel plan takes a worker down
DP_PLANS = [
    ("captured collectives (RCCL inside the update graph)", {}),
    ("eager RCCL collective between two graphs", {"PFRL_GRAPH_COLLECTIVE": "0"}),
    ("process group, host-staged", {"PFRL_RCCL_DIRECT": "0", "PFRL_GRAPH_COLLECTIVE": "0"}),
]


def supervise(args):
    """N > 1: every rank launched by torchrun is a SUPERVISOR that never touches the GPU; the
    workload runs in a child process (this same file, PFRL_BENCH_CHILD=1, its own rendezvous
    port).  A child that dies (a SIGSEGV inside hipStreamEndCapture with RCCL nodes in the graph is
    not an exception anyone can catch -- round 5 met exactly that with a live peer), stalls or
    prints no value costs ONE attempt: the supervisors tell each other through the rendezvous store,
    stop their children, and start the next, more conservative plan of DP_PLANS.  Rank 0 prints the
    first line that every rank completed, with ``config.dp_attempts`` listing what failed before;
    if nothing completes it prints a line with ``value`` null.  Exit status 0 either way: the data
    plane never costs the driver its JSON line."""
    import subprocess

    import torch.distributed as dist

    # (gloo's connection banner goes to fd 1 through C stdio: park fd 1 on stderr, as main() does)
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    base_port = int(os.environ.setdefault("MASTER_PORT", "29500"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    limit = float(os.environ.get("PFRL_BENCH_ATTEMPT_S", "600"))
    attempts, line = [], None
    first = int(os.environ.get("PFRL_BENCH_FIRST_PLAN", "0"))
    for k, (name, extra) in list(enumerate(DP_PLANS))[first:]:
        env = dict(os.environ)
        env.update(extra)
        env.update(PFRL_BENCH_CHILD="1", MASTER_PORT=str(base_port + 101 + k))
        for v in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID"):
            env.pop(v, None)       # (the children rendezvous on a store of their own)
        argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("PFRL_BENCH_CHILD_ARGV"):        # (tests/test_bench_supervisor.py: a stub worker)
            argv = json.loads(os.environ["PFRL_BENCH_CHILD_ARGV"])
        child = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, preexec_fn=_die_with_parent)
        key = "pfrl_bench_attempt_%d_failed" % k
        t0, why, out = time.time(), None, b""
        while True:
            try:
                out, _ = child.communicate(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                if store.add(key, 0) > 0:
                    why = "stopped: a peer's worker failed"
                elif time.time() - t0 > limit:
                    why = "no result within %.0f s" % limit
                if why is not None:
                    child.kill()
                    out, _ = child.communicate()
                    break
        parsed = None
        if why is None and child.returncode != 0:
            why = "worker exited with status %d" % child.returncode
        if why is None and rank == 0:
            try:
                parsed = json.loads(out.decode().strip().splitlines()[-1])
                if parsed.get("value") is None:
                    why = "worker printed no value (%s)" % parsed.get("config", {}).get("dp_plan")
            except Exception as e:      # noqa: BLE001
                why = "worker printed no JSON line (%s)" % (e,)
        if why is not None:
            store.add(key, 1)
        ok = torch.tensor([0.0 if why is not None else 1.0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0.5:
            line = parsed
            break
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        attempts.append({"plan": name, "failed": {str(r): w for r, w in enumerate(reasons) if w}})
        sys.stderr.write("bench.py supervisor: plan '%s' failed (%s)\n" % (name, attempts[-1]["failed"]))
    if rank == 0:
        if line is None:
            line = {"metric": "env-steps/sec whole node (%s)" % args.algo.upper(), "value": None,
                    "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling,
                    "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "not completed", "ranks_seen": world,
                               "dp_plan": "fallback:every data-parallel plan failed"}}
        line.setdefault("config", {})["dp_attempts_failed"] = attempts
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    os.close(result_fd)
    dist.barrier()
    dist.destroy_process_group()


def also_in_own_process(args, argv, limit_s=900, extra_env=None):
    """One ``also`` workload as ``bench.py --algo X`` would measure it ALONE: a process of its own,
    so that nothing an earlier workload of this process left behind (allocator state, captured
    graphs and their pools, module-level hooks of another agent) is part of the number.  (Round 5:
    Rainbow measured 5.8 k env-steps/s as the fifth workload of one process against 9.3-9.9 k alone
    or straight after PPO; the line is about each workload, not about their order.)  Returns the
    child's result dict, or None (the caller then runs the workload in this process)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__)] + argv + [
        "--no-also", "--no-cpu-baseline", "--no-data-path-only", "--seed", str(args.seed)]
    if args.allow_lib_override:
        cmd.append("--allow-lib-override")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PFRL_BENCH_CHILD"):
        env.pop(k, None)
    env.update(extra_env or {})
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env,
                           timeout=limit_s)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stderr.write("bench.py: %s exited with %d; running it in this process\n"
                             % (" ".join(argv), r.returncode))
            return None
        d = json.loads(lines[-1])
        d["config"]["process"] = "its own: " + " ".join(
            ["%s=%s" % kv for kv in sorted((extra_env or {}).items())] + ["bench.py"] + argv)
        return d
    except Exception as e:      # (timeout, unparsable line: the workload still gets measured)
        sys.stderr.write("bench.py: %s in its own process failed (%s); running it in this process\n"
                         % (" ".join(argv), e))
        return None


def rank_shape_legs(args, out, keys):
    """What ONE rank of the 8-GPU job runs, measured on this GPU under a single-rank process group
    (PFRL_DIST_ALWAYS=1: RCCL communicator, data-parallel form of the update with its collectives
    as single-rank launches, the control-plane exchanges): 256 / 8 = 32 envs for DQN, 512 / 8 = 64
    envs x 128 steps with minibatch 2 048 for PPO.  ``projected_speedup_g8`` = 8 x the rank-shaped
    value / the 1-GPU value of the same line: what env sharding gives BEFORE any byte crosses a
    link (the collective's wire time is not in it; DESIGN.md section 6 prices that).  No hardware
    scaling curve is claimed here: the driver measures that itself."""
    dp_env = {"PFRL_DIST_ALWAYS": "1", "PFRL_FORCE_SPLIT_GRAPH": "1", "PFRL_DP_LOWRANK": "force",
              "MASTER_ADDR": "127.0.0.1"}
    legs = (("dqn_rank_shape_g8", out.get("value"),
             ["--algo", "dqn", "--num-envs", "32", "--steps", "160", "--warmup", "40", "--scaling", "weak"]),
            ("ppo_rank_shape_g8", (out["also"].get("ppo") or {}).get("value"),
             ["--algo", "ppo", "--num-envs", "64", "--steps", "128", "--warmup", "128", "--scaling", "weak"]))
    for k, (name, full, argv) in enumerate(legs):
        env = dict(dp_env, MASTER_PORT=str(29731 + k))
        r = also_in_own_process(args, argv, extra_env=env)
        if r is None:
            out["also"][name] = {"value": None, "note": "the rank-shaped run did not complete"}
            continue
        leg = {k_: r[k_] for k_ in keys if k_ in r}
        leg["metric"] = "env-steps/sec of ONE rank of the 8-GPU job (its env shard, data-parallel update)"
        if full:
            leg["projected_speedup_g8"] = round(8.0 * r["value"] / full, 3)
            leg["projection_what"] = ("8 x this value / the 1-GPU value of the full workload on this "
                                      "line; collectives run as single-rank launches, wire time "
                                      "excluded (DESIGN.md section 6)")
        out["also"][name] = leg


def main():
    args = parse_args()
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("PFRL_BENCH_CHILD") != "1"
            and os.environ.get("PFRL_BENCH_SUPERVISE", "1") != "0"):
        return supervise(args)
    # stdout carries exactly ONE line, the JSON result.  Native libraries (RCCL's
    # version banner, MIOpen notes) write to fd 1 through C stdio at times of their
    # own choosing: park fd 1 on stderr for the whole run and keep the real stdout
    # for the result line.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    from pfrl_amd import _native
    from pfrl_amd.distributed import init_process_group_from_env

    rank, world, local = init_process_group_from_env()
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    if world != args.gpus:
        assert world == 1 and args.gpus == 1, "--gpus must equal WORLD_SIZE (use torchrun for N>1)"
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    # the line is about the library the parity tests validated: the in-tree build, nothing else
    assert args.allow_lib_override or not os.environ.get("PFRL_AMD_LIB"), (
        "PFRL_AMD_LIB is set: bench.py measures pfrl_amd/lib/libpfrl_amd.so only (A/B builds of "
        "tools/build_variant.sh: --allow-lib-override, and the line says so)")
    _native.lib()
    if world > 1:
        # one process per GPU: keep each rank's host-side torch ops on its share of cores
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if args.scaling == "strong":
        # the metric's envs are sharded, GPU g owns envs [g N/G, (g+1) N/G)
        assert args.num_envs % world == 0, "--num-envs must divide by the number of GPUs"
        args.total_envs = args.num_envs
        args.num_envs //= world

    if args.blas == "rocblas":
        torch.backends.cuda.preferred_blas_library("cublas")
    elif args.blas == "tunable":
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(10)
        torch.cuda.tunable.set_filename("/tmp/pfrl_tunableop_rank%d.csv" % rank)

    if world > 1:
        _WATCHDOG[0] = _StallWatchdog(args, rank, world, result_fd,
                                      float(os.environ.get("PFRL_BENCH_STALL_S", "300")))
    run_guarded = run_workload
    if torch.distributed.is_initialized():
        from pfrl_amd import rccl

        def run_guarded(*a, **kw):
            """The data plane must not cost the line: an RCCL error code raised on this rank (the
            same call fails on its peers) retires the direct data plane -- the ranks confirm it to
            each other over the control plane -- and the workload is built and run again with the
            gradients carried by the process group (config.dp_plan = "fallback:...")."""
            try:
                return run_workload(*a, **kw)
            except rccl.DataPlaneError as e:
                sys.stderr.write("bench.py: %s -- retrying on the process-group path\n" % (e,))
                why = str(e)[:160]
            torch.distributed.monitored_barrier(timeout=__import__("datetime").timedelta(seconds=120)) \
                if torch.distributed.get_backend() == "gloo" else torch.distributed.barrier()
            rccl.retire(why)
            return run_workload(*a, **kw)

    out = run_guarded(args, device, rank, world)
    if args.algo == "dqn" and not args.host_env and not args.no_also:
        # the other half of BASELINE.json's metric ("DQN 256 envs, PPO 512 envs"): the PPO
        # configs[3] workload, timed by the same process right after
        import copy

        pargs = copy.copy(args)
        pargs.algo, pargs.steps, pargs.warmup = "ppo", 128, 128
        pargs.num_envs = 512 if args.scaling == "weak" else 512 // world
        pargs.cudnn_benchmark = False
        torch.backends.cudnn.benchmark = False
        keys = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "config",
                "roofline")
        own = world == 1 and os.environ.get("PFRL_BENCH_ALSO_IN_PROCESS") != "1"
        torch.cuda.empty_cache()
        also = also_in_own_process(args, ["--algo", "ppo", "--steps", "128", "--warmup", "128",
                                          "--num-envs", "512"]) if own else None
        if also is None:
            also = run_guarded(pargs, device, rank, world, result_extras=False)
        if rank == 0:
            out["also"] = {"ppo": {k: also[k] for k in keys}}
        if world == 1:
            # (also.ppo IS the reference's value pass since round 6 -- PPO(reuse_next_values=False)
            # is the package default; round 5's separate also.ppo_reference_semantics leg is gone)
            # configs[2] and configs[4], short: every GPU config of BASELINE.json on one line
            for algo, n_envs, mb, blas in (("rainbow", 256, 32, "default"), ("sac", 64, 256, "tunable")):
                # (warm-up: every double-buffered minibatch set captures its graphs)
                r2 = also_in_own_process(args, ["--algo", algo, "--steps", "50", "--warmup", "20",
                                                "--num-envs", str(n_envs), "--minibatch", str(mb),
                                                "--blas", blas]) if own else None
                if r2 is None:
                    a2 = copy.copy(args)
                    a2.algo, a2.steps, a2.warmup = algo, 50, 20
                    a2.num_envs, a2.minibatch, a2.blas = n_envs, mb, blas
                    a2.cudnn_benchmark = True
                    torch.backends.cudnn.benchmark = True
                    if blas == "tunable":
                        torch.cuda.tunable.enable(True)
                        torch.cuda.tunable.tuning_enable(True)
                    else:
                        torch.cuda.tunable.enable(False)
                    r2 = run_workload(a2, device, rank, world, result_extras=False)
                if rank == 0:
                    out["also"][algo] = {k: r2[k] for k in keys}
            if own and rank == 0:
                rank_shape_legs(args, out, keys)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.algo == "dqn":
            port = cpu_baseline(args, min(args.cpu_baseline_seconds, 8.0))
            ref = reference_baseline(args)
            if ref is not None:
                ref["port"] = port       # the C oracle + torch-CPU port, as rounds 1-2 reported
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
            if "also" in out and "ppo" in out["also"]:
                # the PPO half of the metric gets its own reference baseline
                pref = reference_baseline_ppo(args)
                if pref is not None:
                    out["also"]["ppo"]["cpu_baseline"] = pref
            for algo, n_envs in (("rainbow", 256), ("sac", 64)):
                if algo in out.get("also", {}):
                    oref = reference_baseline_other(args, algo, n_envs)
                    if oref is not None:
                        out["also"][algo]["cpu_baseline"] = oref
        if not args.no_cpu_baseline and world == 1 and args.algo in ("rainbow", "sac"):
            oref = reference_baseline_other(args, args.algo, args.num_envs)
            if oref is not None:
                out["cpu_baseline"] = oref
        if not args.no_cpu_baseline and world == 1 and args.algo == "ppo":
            pref = reference_baseline_ppo(args, num_envs=args.num_envs)
            if pref is not None:
                out["cpu_baseline"] = pref
    if rank == 0 and out is not None:
        import hashlib

        lib_path = os.environ.get("PFRL_AMD_LIB") or _native.LIB_PATH
        with open(lib_path, "rb") as f:
            out["config"]["native_lib"] = {
                "path": os.path.relpath(lib_path, os.path.dirname(os.path.abspath(__file__))),
                "sha256_16": hashlib.sha256(f.read()).hexdigest()[:16],
                "override": bool(os.environ.get("PFRL_AMD_LIB"))}
    # the line first: teardown of a communicator must not be able to cost it
    sys.stdout.flush()
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if _WATCHDOG[0] is not None:
        _WATCHDOG[0].printed = True
    _tick("teardown")
    if torch.distributed.is_initialized() and os.environ.get("PFRL_BENCH_SOFT_EXIT") != "1":
        # (PFRL_BENCH_SOFT_EXIT=1: a profiler's exit handlers must run -- rocprofv3 writes its
        # trace at exit -- single-rank groups only)
        # The communicator is NOT destroyed: ncclCommDestroy waits for every captured graph that
        # holds one of its collectives to be released first (round 5, two live ranks: the check
        # tool sat in it until its timeout with a graph still referenced), and nothing here needs
        # an orderly RCCL shutdown -- the device is drained, the ranks meet at a barrier, and the
        # process leaves without running finalizers.
        torch.cuda.synchronize()
        try:
            torch.distributed.barrier()
        except Exception:       # noqa: BLE001 -- a peer that already left must not cost the status
            pass
        os.close(result_fd)
        sys.stderr.flush()
        os._exit(0)
    if _WATCHDOG[0] is not None:
        _WATCHDOG[0].done = True
    os.close(result_fd)


if __name__ == "__main__":
    main()
