#!/usr/bin/env python
"""Device pairing of the episodic host containers; also run as ``-m gpu`` tests
(tests/test_episodic_recurrent.py imports the two checks below).

  1. PrioritizedEpisodicReplayBuffer over the HBM priority trees, against the reference traces
     tests/golden/prioritized_episodic_trace_*.npz (the CPU test uses the oracle's tree).
  2. DoubleDQN(recurrent=True) with the model on the GPU (episodes stay on the host), against
     tests/golden/agent_trace_drqn.npz: actions and window lengths exactly, losses loosely.

    gpurun --timeout 600 -- python tools/check_episodic_gpu.py
"""
import glob
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def check_prioritized_episodic(path, payload_on_device=False):
    from pfrl_amd.replay_buffers import PrioritizedEpisodicReplayBuffer

    g = np.load(path)
    seed, cap, _, batch, max_len = (int(v) for v in g["meta"])
    norm = {0: False, 1: True, 2: "memory"}[int(g["normalize"])]
    np.random.seed(seed)
    ur = float(g["uniform_ratio"]) if "uniform_ratio" in g.files else 0
    rbuf = PrioritizedEpisodicReplayBuffer(capacity=None if cap < 0 else cap, betasteps=50,
                                           normalize_by_max=norm, error_max=2.0, uniform_ratio=ur,
                                           device="cuda:0", max_episodes=4096)   # HBM trees
    if payload_on_device:
        rbuf._device_opts = dict(max_size=4096, slack=None, frame_slots=None)
        rbuf.bind("cuda:0")                     # what an agent with gpu >= 0 does: payloads in HBM
        assert rbuf.is_device
    sample_at = {int(k): i for i, k in enumerate(g["s_at_op"])}
    tid = 0
    for k in range(len(g["op_kind"])):
        if g["op_kind"][k] == 1:
            rbuf.stop_current_episode(env_id=int(g["op_env"][k]))
        else:
            obs = (lambda t: np.full(4, t, np.float32)) if payload_on_device else (lambda t: t)
            rbuf.append(state=obs(tid), action=0, reward=0.0, next_state=obs(tid + 1),
                        is_state_terminal=bool(g["op_term"][k]), env_id=int(g["op_env"][k]),
                        tid=tid)
            tid += 1
        left = -1 if rbuf.capacity_left is None else rbuf.capacity_left
        assert (len(rbuf), rbuf.n_episodes, left) == (
            g["length"][k], g["n_episodes"][k], g["cap_left"][k]), k
        if k in sample_at:
            i = sample_at[k]
            sl = slice(i * batch, (i + 1) * batch)
            episodes, weights = rbuf.sample_episodes(batch, max_len=max_len)
            assert [len(ep) for ep in episodes] == list(g["s_ep_len"][sl]), k
            assert [ep[0]["tid"] for ep in episodes] == list(g["s_first_tid"][sl]), k
            np.testing.assert_allclose(weights, g["s_weights"][sl], rtol=1e-6)
            rbuf.update_errors([float(e) for e in g["s_errors"][sl]])
    print("ok  prioritized episodic on device:", os.path.basename(path))


def check_drqn():
    import pfrl_amd
    from pfrl_amd import agents, experiments, explorers
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv
    from pfrl_amd.q_functions import DiscreteActionValueHead
    from pfrl_amd.replay_buffers import EpisodicReplayBuffer

    g = np.load(os.path.join(GOLDEN, "agent_trace_drqn.npz"))
    pfrl_amd.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(4, seed=7, frame_shape=(12, 12), p_done=0.08)
    torch.manual_seed(2468)
    tnn = torch.nn
    q = pfrl_amd.nn.RecurrentSequential(
        tnn.Flatten(), tnn.Linear(4 * 144, 32), tnn.ReLU(), tnn.LSTM(32, 16), tnn.Linear(16, 6),
        DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2)
    rbuf = EpisodicReplayBuffer(300)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 300, lambda: np.random.randint(6))
    ag = agents.DoubleDQN(q, opt, rbuf, 0.99, ex, gpu=0, replay_start_size=40, minibatch_size=4,
                          update_interval=4, target_update_interval=60,
                          phi=lambda x: np.asarray(x, dtype=np.float32) / 255,
                          batch_accumulator="mean", recurrent=True, episodic_update_len=6)
    actions, sampled, losses = [], [], []
    orig_act, orig_update = ag.batch_act, ag.update_from_episodes

    def spy_act(obs):
        a = orig_act(obs)
        actions.append([int(x) for x in a])
        return a

    def spy_update(episodes, errors_out=None):
        sampled.append([len(ep) for ep in episodes])
        orig_update(episodes, errors_out)
        losses.append(float(ag.loss_record.values()[-1]))

    ag.batch_act = spy_act
    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, 480, tempfile.mkdtemp())
    np.testing.assert_array_equal(np.asarray(sampled), g["sampled_len"])
    same = int((np.asarray(actions) == g["actions"]).all(axis=1).sum())
    print("drqn on device: %d / %d steps with identical actions" % (same, len(actions)))
    np.testing.assert_allclose(np.asarray(losses)[:20], g["losses"][:20], rtol=1e-3, atol=1e-5)
    print("ok  drqn on device (first 20 losses within 1e-3)")


if __name__ == "__main__":
    assert torch.cuda.is_available()
    for p in sorted(glob.glob(os.path.join(GOLDEN, "prioritized_episodic_trace_*.npz"))):
        check_prioritized_episodic(p)
    check_drqn()
