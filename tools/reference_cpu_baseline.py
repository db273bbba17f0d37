#!/usr/bin/env python
"""The REFERENCE ITSELF (pfnet/pfrl from /root/reference, unmodified, ``gpu=-1``) on the
benchmark's synthetic workload, timed on the host cores of the BUILD container (SURVEY.md 8d,
"CPU baseline timing").  /root/reference does not exist on the GPU box, so this number cannot
be taken there: it is recorded here, once per round, into profiles/ and bench.py quotes it as
``cpu_baseline.reference_value`` next to the port it times on the GPU box's own cores.

Workload = BASELINE.json configs[1] with the replay capacity cut to 1e5 for host memory:
256 in-process synthetic Atari-shaped envs (VectorFrameStack semantics: LazyFrames of four
84x84 u8 frames, consecutive observations share three frames by identity), DQN with the
Nature CNN exactly as examples/atari/train_dqn_batch_ale.py builds it, ReplayBuffer(1e5),
B = 32, update_interval = 4, RMSprop(centered).  Two figures:
  end_to_end       the agent as is, torch CPU threads = all cores
  data_path_only   the same loop with a zero-FLOP q_function

    python tools/reference_cpu_baseline.py --seconds 40 --out profiles/r02_reference_cpu_baseline.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("PFRL_REFERENCE", "/root/reference")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40.0)
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--capacity", type=int, default=10 ** 5)
    ap.add_argument("--prefill", type=int, default=20000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    sys.path.insert(0, os.path.join(ROOT, "tests", "_gymshim"))   # test-only `gym` stand-in
    sys.path.insert(0, REFERENCE)
    import torch
    import pfrl                                                     # the reference
    from pfrl import agents, explorers, replay_buffers
    from pfrl.q_functions import DiscreteActionValueHead
    from pfrl.wrappers.atari_wrappers import LazyFrames

    assert os.path.realpath(os.path.dirname(pfrl.__file__)).startswith(os.path.realpath(REFERENCE))
    cores = os.cpu_count()
    torch.set_num_threads(cores)
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

    pfrl.utils.set_random_seed(0)
    rbuf = replay_buffers.ReplayBuffer(args.capacity)
    env = SyntheticAtari()
    stub = make_agent(ZeroFlopQ(), rbuf)
    obss = env.reset()
    t0 = time.perf_counter()
    while len(rbuf) < args.prefill:
        obss = one_step(stub, env, obss)
    t_fill = time.perf_counter() - t0

    def measure(agent, obss, seconds):
        steps, t0 = 0, time.perf_counter()
        while True:
            obss = one_step(agent, env, obss)
            steps += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                return steps, el, obss

    s_dp, el_dp, obss = measure(stub, obss, args.seconds * 0.4)
    from pfrl.nn import LargeAtariCNN   # train_dqn_batch_ale.py:35-41, arch "nature"
    from pfrl.initializers import init_chainer_default

    q = torch.nn.Sequential(LargeAtariCNN(), init_chainer_default(torch.nn.Linear(512, n_actions)),
                            DiscreteActionValueHead())
    real = make_agent(q, rbuf)
    real.t = stub.t
    s_e2e, el_e2e, obss = measure(real, obss, args.seconds)
    out = {
        "what": "reference pfnet/pfrl (gpu=-1) on the synthetic configs[1] workload, build container",
        "cores": cores, "torch_threads": cores, "num_envs": N, "capacity": args.capacity,
        "replay_len_at_start": args.prefill, "prefill_s": round(t_fill, 1),
        "end_to_end": {"value": round(s_e2e * N / el_e2e, 2), "unit": "env-steps/s",
                       "steps": s_e2e, "seconds": round(el_e2e, 1),
                       "updates": s_e2e * N // 4},
        "data_path_only": {"value": round(s_dp * N / el_dp, 2), "unit": "env-steps/s",
                           "steps": s_dp, "seconds": round(el_dp, 1),
                           "note": "zero-FLOP q_function: batch_states, append, sample, "
                                   "batch_experiences, loss on a [32, 6] constant"},
        "torch": torch.__version__, "numpy": np.__version__,
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
