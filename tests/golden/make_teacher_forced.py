#!/usr/bin/env python
"""Teacher-forced loss fixtures, recorded by running the REFERENCE (pfnet/pfrl) itself.

    python tests/golden/make_teacher_forced.py      (build container only: needs /root/reference)

The agent traces of make_golden.py compare whole trajectories, so late updates can only be held
to a drift tolerance (2e-4 after ~40 optimizer steps).  This script re-runs the same three
reference agents (same seeds, same environment: the losses it sees are asserted equal to the
committed traces) and records, for updates {1, 50, 140}, everything ONE loss evaluation depends
on -- the online and target parameters BEFORE the update and the minibatch ``exp_batch`` that
``pfrl/agents/dqn.py:407-470`` hands to ``_compute_loss`` -- together with the loss the reference
computed.  A test can then load exactly that state into the device path and compare that one
number at the north-star tolerance (1e-5), wherever in the trajectory it sits.

Output: tests/golden/teacher_forced_{dqn_uniform_n1,ddqn_per_n3,c51_per_n3}.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as mg  # noqa: E402  (puts /root/reference and the gym shim on sys.path)

UPDATES = (1, 50, 140)
KEYS = ("state", "action", "reward", "next_state", "is_state_terminal", "discount", "weights")


def record(ag, out):
    """Wrap the agent's _compute_loss: snapshot (params, target params, exp_batch) before, the
    loss after, at the chosen update numbers (1-based)."""
    orig = ag._compute_loss
    count = [0]

    def spy(exp_batch, errors_out=None):
        count[0] += 1
        k = count[0]
        if k in UPDATES:
            out["u%d_params" % k] = np.concatenate(
                [p.detach().numpy().ravel() for p in ag.model.parameters()])
            out["u%d_target_params" % k] = np.concatenate(
                [p.detach().numpy().ravel() for p in ag.target_model.parameters()])
            for key in KEYS:
                if key in exp_batch:
                    out["u%d_%s" % (k, key)] = exp_batch[key].detach().numpy().copy()
        loss = orig(exp_batch, errors_out)
        if k in UPDATES:
            out["u%d_loss" % k] = np.asarray(float(loss))
        return loss

    ag._compute_loss = spy


def dqn_like(name, prioritized, num_steps, double, steps=640, N=4):
    import tempfile

    import pfrl
    from pfrl import agents, explorers, experiments, replay_buffers
    from pfrl.q_functions import DiscreteActionValueHead

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=3, frame_shape=(12, 12), p_done=0.04)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    q = mg.make_q_function(4 * 144, 6, DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2)
    if prioritized:
        rbuf = replay_buffers.PrioritizedReplayBuffer(
            200, alpha=0.5, beta0=0.4, betasteps=100, num_steps=num_steps,
            normalize_by_max="memory")
    else:
        rbuf = replay_buffers.ReplayBuffer(200, num_steps=num_steps)
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    cls = agents.DoubleDQN if double else agents.DQN
    ag = cls(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40, minibatch_size=8,
             update_interval=4, target_update_interval=60, phi=phi, batch_accumulator="sum")
    out = {}
    record(ag, out)
    losses = []
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        losses.append(ag.loss_record[-1])

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    finish(name, out, losses)


def c51(steps=640, N=4):
    import tempfile

    import pfrl
    from pfrl import agents, explorers, experiments, replay_buffers
    from pfrl.q_functions import DistributionalSingleModelStateQFunctionWithDiscreteAction

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=11, frame_shape=(12, 12), p_done=0.04)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    q = DistributionalSingleModelStateQFunctionWithDiscreteAction(
        mg.DistNet(), np.linspace(-3, 3, 11, dtype=np.float32))
    opt = torch.optim.SGD(q.parameters(), lr=1e-2)
    rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                  num_steps=3, normalize_by_max="memory")
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    ag = agents.CategoricalDoubleDQN(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40,
                                     minibatch_size=8, update_interval=4,
                                     target_update_interval=60, phi=phi, batch_accumulator="mean")
    out = {}
    record(ag, out)
    losses = []
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        losses.append(ag.loss_record[-1])

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    finish("c51_per_n3", out, losses)


def finish(name, out, losses):
    # the same run as the committed trace: same losses, bit for bit
    g = np.load(os.path.join(HERE, "agent_trace_%s.npz" % name))
    assert np.array_equal(np.asarray(losses), g["losses"]), "not the run of agent_trace_%s" % name
    for k in UPDATES:
        assert abs(float(out["u%d_loss" % k]) - losses[k - 1]) <= 1e-12 * max(1.0, abs(losses[k - 1]))
    out["updates"] = np.asarray(UPDATES)
    np.savez_compressed(os.path.join(HERE, "teacher_forced_%s.npz" % name), **out)
    print("teacher_forced", name, {k: float(out["u%d_loss" % k]) for k in UPDATES})


if __name__ == "__main__":
    dqn_like("dqn_uniform_n1", False, 1, False)
    dqn_like("ddqn_per_n3", True, 3, True)
    c51()
