#!/usr/bin/env python
"""The REFERENCE ITSELF (pfnet/pfrl, unmodified, ``gpu=-1``) on the benchmark's synthetic
workload, timed on the host cores of whatever box this runs on ("CPU baseline
timing").  The reference is imported from /root/reference where that exists (the build
container) and otherwise from ``oracle/_ref/`` -- its own modules compiled to sourceless .pyc by
``oracle/build_ref.py``, which travel to the GPU box with the tree -- so ``bench.py`` times it
THERE, next to the MI355X numbers (``cpu_baseline.kind = "reference"``).

Workload = BASELINE.json configs[1] with the replay capacity cut to 1e5 for host memory:
256 in-process synthetic Atari-shaped envs (VectorFrameStack semantics: LazyFrames of four
84x84 u8 frames, consecutive observations share three frames by identity), DQN with the
Nature CNN exactly as examples/atari/train_dqn_batch_ale.py builds it, ReplayBuffer(1e5),
B = 32, update_interval = 4, RMSprop(centered).  Two figures:
  end_to_end       the agent as is, torch CPU threads = all cores
  data_path_only   the same loop with a zero-FLOP q_function

    python tools/reference_cpu_baseline.py --seconds 40 --out profiles/r02_reference_cpu_baseline.json
    python tools/reference_cpu_baseline.py --min-steps 20000 --seeds 0,1,2 --threads all,16
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("PFRL_REFERENCE") or (
    "/root/reference" if os.path.isdir("/root/reference/pfrl") else os.path.join(ROOT, "oracle", "_ref"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40.0, help="end-to-end sample per (seed, threads)")
    ap.add_argument("--dp-seconds", type=float, default=None, help="data-path-only sample (default 0.4 x)")
    ap.add_argument("--min-steps", type=int, default=0,
                    help="keep sampling until this many env-steps as well (BASELINE.md: >= 2e4)")
    ap.add_argument("--seeds", default="0", help="comma list; the median over seeds is reported")
    ap.add_argument("--threads", default="all",
                    help="comma list of torch CPU thread counts for the end-to-end figure ('all' = "
                         "os.cpu_count()); the best one is the headline")
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--capacity", type=int, default=10 ** 5)
    ap.add_argument("--prefill", type=int, default=20000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--algo", choices=["dqn", "ppo", "rainbow", "sac"], default="dqn",
                    help="ppo = BASELINE configs[3]: the reference's PPO (examples/atari/train_ppo_ale.py "
                         "model and hyperparameters), --num-envs envs x --ppo-steps steps per rollout, "
                         "minibatch = rollout / 4, 4 epochs; timed: whole rollouts including their update")
    ap.add_argument("--ppo-steps", type=int, default=128, help="rollout length T (128 = BASELINE's)")
    ap.add_argument("--ppo-rollouts", type=int, default=1, help="timed rollouts (after one untimed)")
    args = ap.parse_args()

    sys.path.insert(0, os.path.join(ROOT, "tests", "_gymshim"))   # test-only `gym` stand-in
    sys.path.insert(0, REFERENCE)
    import torch
    import pfrl                                                     # the reference
    from pfrl import agents, explorers, replay_buffers
    from pfrl.q_functions import DiscreteActionValueHead
    from pfrl.wrappers.atari_wrappers import LazyFrames

    assert os.path.realpath(os.path.dirname(pfrl.__file__)).startswith(os.path.realpath(REFERENCE)), \
        "not the reference: %s" % pfrl.__file__
    cores = os.cpu_count()
    thread_counts = [cores if t == "all" else min(cores, int(t)) for t in args.threads.split(",")]
    torch.set_num_threads(thread_counts[0])
    N, n_actions = args.num_envs, 6

    class SyntheticAtari(pfrl.env.VectorEnv):
        """iid U{0..255} frames, reward in {-1, 0, 1}, done w.p. 1/500."""

        def __init__(self):
            self.num_envs = N
            self.rs = [np.random.RandomState(1000 + i) for i in range(N)]
            self.stacks = [None] * N

        def _frame(self, i):
            return self.rs[i].randint(0, 256, size=(1, 84, 84)).astype(np.uint8)

        def _obs(self):
            return [LazyFrames(list(s), stack_axis=0) for s in self.stacks]

        def reset(self, mask=None):
            for i in range(N):
                if mask is None or not mask[i]:
                    f = self._frame(i)
                    self.stacks[i] = [f, f, f, f]
            return self._obs()

        def step(self, actions):
            rews, dones = [], []
            for i in range(N):
                self.stacks[i] = self.stacks[i][1:] + [self._frame(i)]
                u = self.rs[i].rand()
                rews.append(-1.0 if u < 0.05 else (1.0 if u > 0.95 else 0.0))
                dones.append(bool(self.rs[i].rand() < 1.0 / 500))
            return self._obs(), rews, dones, [{} for _ in range(N)]

        def seed(self, seeds=None):
            pass

        def close(self):
            pass

    def phi(x):   # examples/atari/train_dqn_batch_ale.py:229-231
        return np.asarray(x, dtype=np.float32) / 255

    if args.algo == "ppo":
        return ppo_baseline(args, pfrl, torch, SyntheticAtari, phi, N, n_actions, cores, thread_counts)
    if args.algo == "rainbow":
        return rainbow_baseline(args, pfrl, torch, SyntheticAtari, phi, N, n_actions, cores, thread_counts)
    if args.algo == "sac":
        return sac_baseline(args, pfrl, torch, N, cores, thread_counts)

    class ZeroFlopQ(torch.nn.Module):
        """Q-values that do not depend on the observation: one learnable row."""

        def __init__(self):
            super().__init__()
            self.q = torch.nn.Parameter(torch.zeros(1, n_actions))

        def forward(self, x):
            return pfrl.action_value.DiscreteActionValue(self.q.expand(x.shape[0], n_actions))

    def make_agent(q_func, rbuf):
        # examples/atari/train_dqn_batch_ale.py:199-206
        opt = torch.optim.RMSprop(q_func.parameters(), lr=2.5e-4, alpha=0.95, momentum=0.0,
                                  eps=1e-2, centered=True)
        explorer = explorers.LinearDecayEpsilonGreedy(1.0, 0.01, 10 ** 6,
                                                      lambda: np.random.randint(n_actions))
        return agents.DQN(q_func, opt, rbuf, gpu=-1, gamma=0.99, explorer=explorer,
                          replay_start_size=args.prefill, target_update_interval=3 * 10 ** 4,
                          clip_delta=True, update_interval=4, minibatch_size=32,
                          batch_accumulator="sum", phi=phi)

    def one_step(agent, env, obss):
        actions = agent.batch_act(obss)
        obss, rs, dones, infos = env.step(actions)
        agent.batch_observe(obss, rs, dones, [False] * N)
        return env.reset([not d for d in dones])

    from pfrl.nn import LargeAtariCNN   # train_dqn_batch_ale.py:35-41, arch "nature"
    from pfrl.initializers import init_chainer_default

    def measure(agent, env, obss, seconds, min_steps):
        steps, t0 = 0, time.perf_counter()
        while True:
            obss = one_step(agent, env, obss)
            steps += 1
            el = time.perf_counter() - t0
            if el >= seconds and steps * N >= min_steps:
                return steps, el, obss

    dp_seconds = args.dp_seconds if args.dp_seconds is not None else args.seconds * 0.4
    runs = []
    for seed in [int(x) for x in args.seeds.split(",")]:
        pfrl.utils.set_random_seed(seed)
        rbuf = replay_buffers.ReplayBuffer(args.capacity)
        env = SyntheticAtari()
        stub = make_agent(ZeroFlopQ(), rbuf)
        obss = env.reset()
        t0 = time.perf_counter()
        while len(rbuf) < args.prefill:
            obss = one_step(stub, env, obss)
        t_fill = time.perf_counter() - t0
        s_dp, el_dp, obss = measure(stub, env, obss, dp_seconds, args.min_steps)
        run = {"seed": seed, "prefill_s": round(t_fill, 1),
               "data_path_only": {"value": round(s_dp * N / el_dp, 2), "steps": s_dp,
                                  "seconds": round(el_dp, 1)},
               "end_to_end": {}}
        q = torch.nn.Sequential(LargeAtariCNN(),
                                init_chainer_default(torch.nn.Linear(512, n_actions)),
                                DiscreteActionValueHead())
        real = make_agent(q, rbuf)
        real.t = stub.t
        for th in thread_counts:
            torch.set_num_threads(th)
            s_e2e, el_e2e, obss = measure(real, env, obss, args.seconds, args.min_steps)
            run["end_to_end"][str(th)] = {"value": round(s_e2e * N / el_e2e, 2), "steps": s_e2e,
                                          "seconds": round(el_e2e, 1), "updates": s_e2e * N // 4}
        runs.append(run)

    def median(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])

    by_threads = {str(th): median([r["end_to_end"][str(th)]["value"] for r in runs])
                  for th in thread_counts}
    best = max(by_threads, key=lambda k: by_threads[k])
    out = {
        "what": "reference pfnet/pfrl (gpu=-1) on the synthetic configs[1] workload",
        "reference_from": REFERENCE, "host_cores": cores, "torch_threads": int(best),
        "cores": int(best), "num_envs": N, "capacity": args.capacity,
        "replay_len_at_start": args.prefill, "seeds": [r["seed"] for r in runs],
        "end_to_end": {"value": round(by_threads[best], 2), "unit": "env-steps/s",
                       "median_over_seeds_by_threads": by_threads,
                       "env_steps_per_sample": [r["end_to_end"][best]["steps"] * N for r in runs]},
        "data_path_only": {"value": round(median([r["data_path_only"]["value"] for r in runs]), 2),
                           "unit": "env-steps/s",
                           "env_steps_per_sample": [r["data_path_only"]["steps"] * N for r in runs],
                           "note": "zero-FLOP q_function: batch_states, append, sample, "
                                   "batch_experiences, loss on a [32, 6] constant"},
        "runs": runs,
        "torch": torch.__version__, "numpy": np.__version__,
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


def ppo_baseline(args, pfrl, torch, SyntheticAtari, phi, N, n_actions, cores, thread_counts):
    """The reference's PPO (pfrl/agents/ppo.py, gpu=-1) as examples/atari/train_ppo_ale.py:247-264
    builds it, on N in-process synthetic envs: whole rollouts of T steps INCLUDING the update each
    one ends with (value pass, GAE, 4 epochs of minibatches); env-steps/s = N * T / wall."""
    from pfrl import agents
    from pfrl.policies import SoftmaxCategoricalHead

    def lecun_init(layer, gain=1):
        pfrl.initializers.init_lecun_normal(layer.weight, gain)
        torch.nn.init.zeros_(layer.bias)
        return layer

    nn = torch.nn
    T = args.ppo_steps
    th = thread_counts[0]
    torch.set_num_threads(th)
    pfrl.utils.set_random_seed(int(args.seeds.split(",")[0]))
    model = nn.Sequential(
        lecun_init(nn.Conv2d(4, 32, 8, stride=4)), nn.ReLU(),
        lecun_init(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
        lecun_init(nn.Conv2d(64, 64, 3, stride=1)), nn.ReLU(), nn.Flatten(),
        lecun_init(nn.Linear(3136, 512)), nn.ReLU(),
        pfrl.nn.Branched(
            nn.Sequential(lecun_init(nn.Linear(512, n_actions), 1e-2), SoftmaxCategoricalHead()),
            lecun_init(nn.Linear(512, 1))))
    opt = torch.optim.Adam(model.parameters(), lr=2.5e-4, eps=1e-5)
    agent = agents.PPO(model, opt, gpu=-1, phi=phi, update_interval=N * T, minibatch_size=N * T // 4,
                       epochs=4, clip_eps=0.1, clip_eps_vf=None, standardize_advantages=True,
                       entropy_coef=1e-2, max_grad_norm=0.5)
    env = SyntheticAtari()
    obss = env.reset()

    def rollout(obss):
        n0 = agent.n_updates
        for _ in range(T):
            actions = agent.batch_act(obss)
            obss, rs, dones, infos = env.step(actions)
            agent.batch_observe(obss, rs, dones, [False] * N)
            obss = env.reset([not d for d in dones])
        assert agent.n_updates > n0, "the rollout did not end with an update"
        return obss

    t0 = time.perf_counter()
    obss = rollout(obss)                       # untimed: allocator / thread pool warm-up
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.ppo_rollouts):
        obss = rollout(obss)
    el = time.perf_counter() - t0
    out = {
        "what": "reference pfnet/pfrl PPO (gpu=-1) on the synthetic configs[3] workload",
        "reference_from": REFERENCE, "host_cores": cores, "torch_threads": th, "cores": th,
        "num_envs": N, "rollout_steps": T, "update_interval": N * T, "minibatch": N * T // 4,
        "epochs": 4, "rollouts_timed": args.ppo_rollouts, "warmup_rollout_s": round(t_warm, 1),
        "end_to_end": {"value": round(args.ppo_rollouts * N * T / el, 2), "unit": "env-steps/s",
                       "seconds": round(el, 1), "env_steps": args.ppo_rollouts * N * T,
                       "updates": agent.n_updates},
        "torch": torch.__version__, "numpy": np.__version__,
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


def _timed_steps(one_step, obss, seconds, N):
    steps, t0 = 0, time.perf_counter()
    while True:
        obss = one_step(obss)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            return steps, el, obss


def rainbow_baseline(args, pfrl, torch, SyntheticAtari, phi, N, n_actions, cores, thread_counts):
    """The reference's Rainbow (examples/atari/reproduction/rainbow/train_rainbow.py:110-159, gpu=-1):
    CategoricalDoubleDQN on DistributionalDuelingDQN(51 atoms) with factorised NoisyNet,
    PrioritizedReplayBuffer(alpha 0.5, beta0 0.4, num_steps 3, normalize_by_max='memory'),
    Adam(6.25e-5, eps 1.5e-4), B = 32, update_interval 4 -- BASELINE configs[2] on N in-process
    synthetic envs, capacity cut to --capacity for host memory."""
    from pfrl import agents, explorers, replay_buffers
    from pfrl.q_functions import DistributionalDuelingDQN

    th = thread_counts[0]
    torch.set_num_threads(th)
    pfrl.utils.set_random_seed(int(args.seeds.split(",")[0]))
    q_func = DistributionalDuelingDQN(n_actions, 51, -10, 10)
    pfrl.nn.to_factorized_noisy(q_func, sigma_scale=0.5)
    opt = torch.optim.Adam(q_func.parameters(), 6.25e-5, eps=1.5 * 10 ** -4)
    rbuf = replay_buffers.PrioritizedReplayBuffer(args.capacity, alpha=0.5, beta0=0.4,
                                                  betasteps=2 * 10 ** 6, num_steps=3,
                                                  normalize_by_max="memory")
    agent = agents.CategoricalDoubleDQN(
        q_func, opt, rbuf, gpu=-1, gamma=0.99, explorer=explorers.Greedy(), minibatch_size=32,
        replay_start_size=args.capacity, target_update_interval=32000, update_interval=4,
        batch_accumulator="mean", phi=phi)
    env = SyntheticAtari()
    obss = env.reset()

    def one_step(obss):
        actions = agent.batch_act(obss)
        obss, rs, dones, infos = env.step(actions)
        agent.batch_observe(obss, rs, dones, [False] * N)
        return env.reset([not d for d in dones])

    t0 = time.perf_counter()
    while len(rbuf) < args.prefill:
        obss = one_step(obss)
    t_fill = time.perf_counter() - t0
    agent.replay_updater.replay_start_size = args.prefill
    obss = one_step(obss)                       # untimed: first updates
    u0 = agent.optim_t
    steps, el, obss = _timed_steps(one_step, obss, args.seconds, N)
    out = {
        "what": "reference pfnet/pfrl Rainbow (gpu=-1) on the synthetic configs[2] workload",
        "reference_from": REFERENCE, "host_cores": cores, "torch_threads": th, "cores": th,
        "num_envs": N, "capacity": args.capacity, "replay_len_at_start": args.prefill,
        "prefill_s": round(t_fill, 1),
        "end_to_end": {"value": round(steps * N / el, 2), "unit": "env-steps/s",
                       "seconds": round(el, 1), "env_steps": steps * N,
                       "updates": agent.optim_t - u0},
        "torch": torch.__version__, "numpy": np.__version__,
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


def sac_baseline(args, pfrl, torch, N, cores, thread_counts):
    """The reference's SAC (examples/mujoco/reproduction/soft_actor_critic/
    train_soft_actor_critic.py:172-243, gpu=-1): 256-256 MLP squashed-Gaussian policy and twin Q,
    Adam(3e-4), B = 256, update_interval 1, learned temperature -- BASELINE configs[4] on N
    in-process Humanoid-shaped synthetic envs (obs f32[376] ~ N(0,1), action f32[17], reward
    ~ N(0,1), done w.p. 1/1000), capacity cut to --capacity for host memory."""
    from pfrl import agents, replay_buffers
    from torch import distributions as D
    from torch import nn

    obs_size, action_size = 376, 17
    th = thread_counts[0]
    torch.set_num_threads(th)
    pfrl.utils.set_random_seed(int(args.seeds.split(",")[0]))

    class SyntheticVectorObs(pfrl.env.VectorEnv):
        def __init__(self):
            self.num_envs = N
            self.rs = np.random.RandomState(1000)

        def _obs(self):
            return list(self.rs.randn(N, obs_size).astype(np.float32))

        def reset(self, mask=None):
            return self._obs()

        def step(self, actions):
            rews = [float(r) for r in self.rs.randn(N)]
            dones = [bool(d) for d in self.rs.rand(N) < 1.0 / 1000]
            return self._obs(), rews, dones, [{} for _ in range(N)]

        def seed(self, seeds=None):
            pass

        def close(self):
            pass

    def squashed_diagonal_gaussian_head(x):
        mean, log_scale = torch.chunk(x, 2, dim=1)
        var = torch.exp(torch.clamp(log_scale, -20.0, 2.0) * 2)
        base = D.Independent(D.Normal(loc=mean, scale=torch.sqrt(var)), 1)
        return D.transformed_distribution.TransformedDistribution(
            base, [D.transforms.TanhTransform(cache_size=1)])

    policy = nn.Sequential(nn.Linear(obs_size, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                           nn.Linear(256, action_size * 2), pfrl.nn.lmbda.Lambda(squashed_diagonal_gaussian_head))
    for i in (0, 2, 4):
        nn.init.xavier_uniform_(policy[i].weight)

    def make_q():
        q = nn.Sequential(pfrl.nn.ConcatObsAndAction(), nn.Linear(obs_size + action_size, 256),
                          nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1))
        for i in (1, 3, 5):
            nn.init.xavier_uniform_(q[i].weight)
        return q, torch.optim.Adam(q.parameters(), lr=3e-4)

    q1, q1opt = make_q()
    q2, q2opt = make_q()
    rbuf = replay_buffers.ReplayBuffer(args.capacity)
    agent = agents.SoftActorCritic(
        policy, q1, q2, torch.optim.Adam(policy.parameters(), lr=3e-4), q1opt, q2opt, rbuf,
        gamma=0.99, gpu=-1, replay_start_size=1 << 62, minibatch_size=256, update_interval=1,
        burnin_action_func=lambda: np.random.uniform(-1, 1, size=action_size).astype(np.float32),
        entropy_target=-action_size, temperature_optimizer_lr=3e-4)
    env = SyntheticVectorObs()
    obss = env.reset()

    def one_step(obss):
        actions = agent.batch_act(obss)
        obss, rs, dones, infos = env.step(actions)
        agent.batch_observe(obss, rs, dones, [False] * N)
        return obss

    t0 = time.perf_counter()
    while len(rbuf) < args.prefill:
        obss = one_step(obss)
    t_fill = time.perf_counter() - t0
    agent.replay_start_size = args.prefill
    agent.replay_updater.replay_start_size = args.prefill
    obss = one_step(obss)
    u0 = agent.n_policy_updates
    steps, el, obss = _timed_steps(one_step, obss, args.seconds, N)
    out = {
        "what": "reference pfnet/pfrl SoftActorCritic (gpu=-1) on the synthetic configs[4] workload",
        "reference_from": REFERENCE, "host_cores": cores, "torch_threads": th, "cores": th,
        "num_envs": N, "capacity": args.capacity, "replay_len_at_start": args.prefill,
        "prefill_s": round(t_fill, 1),
        "end_to_end": {"value": round(steps * N / el, 2), "unit": "env-steps/s",
                       "seconds": round(el, 1), "env_steps": steps * N,
                       "updates": agent.n_policy_updates - u0},
        "torch": torch.__version__, "numpy": np.__version__,
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
