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

Output: tests/golden/teacher_forced_{dqn_uniform_n1,ddqn_per_n3,c51_per_n3,ppo,sac,td3,ddpg,iqn_per_n3,a2c}.npz
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


PPO_UPDATES = (1, 9, 20, 32)        # minibatch updates of agent_trace_ppo (8 per rollout)


def ppo(steps=280, N=4):
    """The reference's PPO run of agent_trace_ppo.npz again (asserted: same losses), recording for
    minibatch updates PPO_UPDATES what ONE evaluation of pfrl/agents/ppo.py:480-532 depends on:
    the model's parameters before the step, the minibatch's states (after phi), actions and the
    four columns _lossfun reads (standardised advantages, old log-probabilities, old values, value
    targets) -- and the three numbers it produced (loss, value loss, policy loss)."""
    import tempfile

    import pfrl
    from pfrl import agents, experiments
    from pfrl.policies import SoftmaxCategoricalHead

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=5, frame_shape=(12, 12), p_done=0.06)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    model = mg.make_ppo_model(4 * 144, 6, SoftmaxCategoricalHead, pfrl.nn.Branched)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    ag = agents.PPO(model, opt, gpu=-1, gamma=0.99, lambd=0.95, phi=phi, update_interval=64,
                    minibatch_size=16, epochs=2, clip_eps=0.1, clip_eps_vf=None,
                    standardize_advantages=True, max_grad_norm=0.5)
    out, losses = {}, []
    count = [0]
    last_states = [None]
    orig_forward = model.forward

    def spy_forward(x):
        last_states[0] = x
        return orig_forward(x)

    model.forward = spy_forward
    orig_loss = ag._lossfun

    def spy_loss(entropy, vs_pred, log_probs, vs_pred_old, log_probs_old, advs, vs_teacher):
        count[0] += 1
        k = count[0]
        loss = orig_loss(entropy, vs_pred, log_probs, vs_pred_old=vs_pred_old,
                         log_probs_old=log_probs_old, advs=advs, vs_teacher=vs_teacher)
        losses.append([float(loss), ag.value_loss_record[-1], ag.policy_loss_record[-1]])
        if k in PPO_UPDATES:
            states = last_states[0]
            distribs, _ = orig_forward(states)
            # the actions of the minibatch: the ones whose log-probability _lossfun was given
            lp_all = distribs.logits - distribs.logits.logsumexp(-1, keepdim=True)
            actions = (lp_all - log_probs[:, None]).abs().argmin(-1)
            assert torch.equal(distribs.log_prob(actions), log_probs)
            out["u%d_params" % k] = np.concatenate([p.detach().numpy().ravel() for p in model.parameters()])
            out["u%d_states" % k] = states.detach().numpy().copy()
            out["u%d_actions" % k] = actions.numpy().copy()
            for name, t in (("advs", advs), ("log_probs_old", log_probs_old),
                            ("vs_pred_old", vs_pred_old), ("vs_teacher", vs_teacher)):
                out["u%d_%s" % (k, name)] = t.detach().numpy().copy()
            out["u%d_losses" % k] = np.asarray(losses[-1])
        return loss

    ag._lossfun = spy_loss
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    g = np.load(os.path.join(HERE, "agent_trace_ppo.npz"))
    assert np.array_equal(np.asarray(losses), g["losses"]), "not the run of agent_trace_ppo"
    out["updates"] = np.asarray(PPO_UPDATES)
    out["hyper"] = np.asarray([0.1, 1.0, 0.01])     # clip_eps, value_func_coef, entropy_coef
    np.savez_compressed(os.path.join(HERE, "teacher_forced_ppo.npz"), **out)
    print("teacher_forced ppo", {k: out["u%d_losses" % k].tolist() for k in PPO_UPDATES})


SAC_UPDATES = (1, 90, 180)


def sac(steps=240, N=2, obs_dim=24, act_dim=3):
    """The reference's SoftActorCritic run of agent_trace_sac.npz again (asserted: same Q losses),
    recording for updates SAC_UPDATES what one ``update`` (pfrl/agents/soft_actor_critic.py:
    214-300) depends on -- the five networks' parameters before it and the minibatch -- and its
    three losses: the two critic losses and the policy loss (evaluated, as the reference does,
    after the critics' steps).  Sampling noise is off on both sides (sample = mean), as in the
    trace: the CPU and GPU generators differ."""
    import tempfile

    import pfrl
    from pfrl import agents, experiments, replay_buffers

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=2, p_done=0.03)
    policy, q1, q2 = mg.make_sac_nets(obs_dim, act_dim, pfrl.nn.ConcatObsAndAction, pfrl.nn.Lambda)
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
    rbuf = replay_buffers.ReplayBuffer(500)
    ag = agents.SoftActorCritic(
        policy, q1, q2, opts[0], opts[1], opts[2], rbuf, gamma=0.99, gpu=-1,
        replay_start_size=40, minibatch_size=16, update_interval=1,
        burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
        entropy_target=None, initial_temperature=0.2, soft_update_tau=5e-3)
    flat = lambda m: np.concatenate([p.detach().numpy().ravel() for p in m.parameters()])  # noqa: E731
    out, q_losses = {}, []
    count = [0]
    orig_q, orig_p = ag.update_q_func, ag.update_policy_and_temperature

    def spy_q(batch):
        count[0] += 1
        k = count[0]
        if k in SAC_UPDATES:
            for name, m in (("policy", policy), ("q1", q1), ("q2", q2), ("tq1", ag.target_q_func1),
                            ("tq2", ag.target_q_func2)):
                out["u%d_%s_params" % (k, name)] = flat(m)
            for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount"):
                out["u%d_%s" % (k, key)] = batch[key].detach().numpy().copy()
        orig_q(batch)
        q_losses.append([ag.q_func1_loss_record[-1], ag.q_func2_loss_record[-1]])
        if k in SAC_UPDATES:
            out["u%d_q_losses" % k] = np.asarray(q_losses[-1])

    def spy_p(batch):
        k = count[0]
        if k in SAC_UPDATES:
            with torch.no_grad():       # :273-286 on the parameters as they are now
                s0 = batch["state"]
                d = policy(s0)
                a = d.rsample()
                q = torch.min(q1((s0, a)), q2((s0, a)))
                out["u%d_policy_loss" % k] = np.asarray(float(torch.mean(ag.temperature * d.log_prob(a)[..., None] - q)))
        orig_p(batch)

    ag.update_q_func, ag.update_policy_and_temperature = spy_q, spy_p
    with mg._NoNoise():
        experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    g = np.load(os.path.join(HERE, "agent_trace_sac.npz"))
    assert np.array_equal(np.asarray(q_losses), g["q_losses"]), "not the run of agent_trace_sac"
    out["updates"] = np.asarray(SAC_UPDATES)
    np.savez_compressed(os.path.join(HERE, "teacher_forced_sac.npz"), **out)
    print("teacher_forced sac", {k: (out["u%d_q_losses" % k].tolist(), float(out["u%d_policy_loss" % k]))
                                 for k in SAC_UPDATES})


TD3_UPDATES = (1, 90, 180)


def td3(steps=260, N=2, obs_dim=24, act_dim=3):
    """The reference's TD3 run of agent_trace_td3.npz again (asserted: same critic losses),
    recording for updates TD3_UPDATES what one ``update`` (pfrl/agents/td3.py:181-262) depends on --
    the six networks' parameters before it (policy, twin Q and their three targets) and the
    minibatch -- and its losses: both critic losses, and for updates that also step the policy
    (every ``policy_update_delay`` = 2nd) the policy loss, evaluated as the reference does after
    the critics' steps.  Exploration noise and target smoothing are the deterministic stand-ins of
    the trace (the CPU and GPU generators differ by construction)."""
    import tempfile

    import pfrl
    from pfrl import agents, experiments, explorers, replay_buffers

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=4, p_done=0.03)
    torch.manual_seed(2468)
    policy, q = mg._det_nets(obs_dim, act_dim, pfrl.nn, pfrl.policies)
    q1, q2 = q(), q()
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
    ag = agents.TD3(policy, q1, q2, opts[0], opts[1], opts[2], replay_buffers.ReplayBuffer(500),
                    gamma=0.99, explorer=explorers.AdditiveGaussian(scale=0.1, low=-1.0, high=1.0),
                    gpu=-1, replay_start_size=40, minibatch_size=16, update_interval=1,
                    soft_update_tau=5e-3,
                    burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
                    policy_update_delay=2, target_policy_smoothing_func=mg._shifted_smoothing)
    flat = lambda m: np.concatenate([p.detach().numpy().ravel() for p in m.parameters()])  # noqa: E731
    out, q_losses = {}, []
    count = [0]
    orig_q, orig_p = ag.update_q_func, ag.update_policy

    def spy_q(batch):
        count[0] += 1
        k = count[0]
        if k in TD3_UPDATES:
            for name, m in (("policy", policy), ("q1", q1), ("q2", q2), ("tpolicy", ag.target_policy),
                            ("tq1", ag.target_q_func1), ("tq2", ag.target_q_func2)):
                out["u%d_%s_params" % (k, name)] = flat(m)
            for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount"):
                out["u%d_%s" % (k, key)] = batch[key].detach().numpy().copy()
        orig_q(batch)
        q_losses.append([ag.q_func1_loss_record[-1], ag.q_func2_loss_record[-1]])
        if k in TD3_UPDATES:
            out["u%d_q_losses" % k] = np.asarray(q_losses[-1])
            out["u%d_with_policy" % k] = np.asarray(int(ag.q_func_n_updates % ag.policy_update_delay == 0))

    def spy_p(batch):
        k = count[0]
        orig_p(batch)
        if k in TD3_UPDATES:
            out["u%d_policy_loss" % k] = np.asarray(float(ag.policy_loss_record[-1]))

    ag.update_q_func, ag.update_policy = spy_q, spy_p
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    g = np.load(os.path.join(HERE, "agent_trace_td3.npz"))
    assert np.array_equal(np.asarray(q_losses), g["losses"]), "not the run of agent_trace_td3"
    out["updates"] = np.asarray(TD3_UPDATES)
    np.savez_compressed(os.path.join(HERE, "teacher_forced_td3.npz"), **out)
    print("teacher_forced td3", {k: (out["u%d_q_losses" % k].tolist(), int(out["u%d_with_policy" % k]),
                                     float(out.get("u%d_policy_loss" % k, np.nan))) for k in TD3_UPDATES})


def ddpg(steps=260, N=2, obs_dim=24, act_dim=3):
    """The reference's DDPG run of agent_trace_ddpg.npz again (asserted: same losses), recording
    for updates TD3_UPDATES what one ``update`` (pfrl/agents/ddpg.py:150-200) depends on -- policy,
    Q-function and their two targets before it, and the minibatch -- and both losses: the critic's,
    and the actor's evaluated as the reference does, AFTER the critic's step."""
    import tempfile

    import pfrl
    from pfrl import agents, experiments, explorers, replay_buffers

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=4, p_done=0.03)
    torch.manual_seed(2468)
    policy, q = mg._det_nets(obs_dim, act_dim, pfrl.nn, pfrl.policies)
    q1 = q()
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1)]
    ag = agents.DDPG(policy, q1, opts[0], opts[1], replay_buffers.ReplayBuffer(500), gamma=0.99,
                     explorer=explorers.AdditiveGaussian(scale=0.1, low=-1.0, high=1.0), gpu=-1,
                     replay_start_size=40, minibatch_size=16, update_interval=1,
                     target_update_interval=7, target_update_method="soft", soft_update_tau=5e-2,
                     burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32))
    flat = lambda m: np.concatenate([p.detach().numpy().ravel() for p in m.parameters()])  # noqa: E731
    out, losses = {}, []
    count = [0]
    orig_c, orig_a = ag.compute_critic_loss, ag.compute_actor_loss

    def spy_c(batch):
        count[0] += 1
        k = count[0]
        if k in TD3_UPDATES:
            for name, m in (("policy", policy), ("q", q1), ("tpolicy", ag.target_policy),
                            ("tq", ag.target_q_function)):
                out["u%d_%s_params" % (k, name)] = flat(m)
            for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount"):
                out["u%d_%s" % (k, key)] = batch[key].detach().numpy().copy()
        loss = orig_c(batch)
        if k in TD3_UPDATES:
            out["u%d_critic_loss" % k] = np.asarray(float(loss))
        return loss

    def spy_a(batch):
        k = count[0]
        loss = orig_a(batch)
        if k in TD3_UPDATES:
            out["u%d_actor_loss" % k] = np.asarray(float(loss))
        losses.append([ag.critic_loss_record[-1], ag.actor_loss_record[-1]])
        return loss

    ag.compute_critic_loss, ag.compute_actor_loss = spy_c, spy_a
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    g = np.load(os.path.join(HERE, "agent_trace_ddpg.npz"))
    assert np.array_equal(np.asarray(losses), g["losses"]), "not the run of agent_trace_ddpg"
    out["updates"] = np.asarray(TD3_UPDATES)
    np.savez_compressed(os.path.join(HERE, "teacher_forced_ddpg.npz"), **out)
    print("teacher_forced ddpg", {k: (float(out["u%d_critic_loss" % k]), float(out["u%d_actor_loss" % k]))
                                  for k in TD3_UPDATES})


A2C_UPDATES = {True: (1, 12, 30), False: (1, 30)}


def _as_u8(x):
    """phi(x) = float32(x) / 255 of u8 frames, stored as the frames (a quarter of the bytes, and
    they compress): float32(u8) / 255 gives every value back exactly (asserted)."""
    u = np.rint(np.asarray(x, dtype=np.float64) * 255).astype(np.uint8)
    assert np.array_equal(u.astype(np.float32) / 255, x)
    return u


def a2c(steps=600, N=4):
    """The reference's A2C (the networks / environment of agent_trace_a2c_gae{0,1}, a longer run:
    30 updates) recording, for the updates of A2C_UPDATES, what ONE ``update``
    (pfrl/agents/a2c.py:169-213) depends on -- the parameters before it and the rollout storage
    (states, actions, rewards, masks, value predictions) -- and what it produces: the three loss
    terms (read through the moving averages with their decay set to 0 for that update, which makes
    the average the term itself) and the parameters after the clipped SGD step."""
    import tempfile

    import pfrl
    from pfrl import agents, experiments
    from pfrl.policies import SoftmaxCategoricalHead

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    out = {}
    for use_gae in (True, False):
        pfrl.utils.set_random_seed(0)
        env = HostSyntheticAtariVectorEnv(N, seed=7, frame_shape=(12, 12), p_done=0.08)

        def phi(x):
            return np.asarray(x, dtype=np.float32) / 255

        model = mg.make_ppo_model(4 * 144, 6, SoftmaxCategoricalHead, pfrl.nn.Branched)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        ag = agents.A2C(model, opt, gamma=0.99, num_processes=N, gpu=-1, update_steps=5, phi=phi,
                        use_gae=use_gae, tau=0.95, max_grad_norm=0.5)
        flat = lambda: np.concatenate([p.detach().numpy().ravel() for p in model.parameters()])  # noqa: E731
        count = [0]
        orig = ag.update
        tag = "gae%d" % int(use_gae)

        def spy(_orig=orig, _ag=ag, _count=count, _tag=tag, _use_gae=use_gae, _flat=flat):
            _count[0] += 1
            k = _count[0]
            if k not in A2C_UPDATES[_use_gae]:
                return _orig()
            pre = "%s_u%d_" % (_tag, k)
            out[pre + "params"] = _flat()
            for key in ("states", "actions", "rewards", "masks", "value_preds"):
                out[pre + key] = getattr(_ag, key).detach().numpy().copy()
            out[pre + "states"] = _as_u8(out[pre + "states"])
            saved = (_ag.average_actor_loss, _ag.average_value, _ag.average_entropy,
                     _ag.average_actor_loss_decay, _ag.average_value_decay, _ag.average_entropy_decay)
            _ag.average_actor_loss_decay = _ag.average_value_decay = _ag.average_entropy_decay = 0.0
            _ag.average_actor_loss = _ag.average_value = _ag.average_entropy = 0.0
            _orig()
            out[pre + "losses"] = np.asarray([_ag.average_value, _ag.average_actor_loss,
                                              _ag.average_entropy])        # value, action, entropy
            (_, _, _, _ag.average_actor_loss_decay, _ag.average_value_decay,
             _ag.average_entropy_decay) = saved
            # the moving averages as the unmodified run would hold them
            _ag.average_actor_loss = saved[0] + (1 - saved[3]) * (out[pre + "losses"][1] - saved[0])
            _ag.average_value = saved[1] + (1 - saved[4]) * (out[pre + "losses"][0] - saved[1])
            _ag.average_entropy = saved[2] + (1 - saved[5]) * (out[pre + "losses"][2] - saved[2])
            out[pre + "returns"] = _ag.returns.numpy().copy()
            out[pre + "params_after"] = _flat()

        ag.update = spy
        experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
        assert count[0] >= max(A2C_UPDATES[use_gae]), count
        out[tag + "_updates"] = np.asarray(A2C_UPDATES[use_gae])
    np.savez_compressed(os.path.join(HERE, "teacher_forced_a2c.npz"), **out)
    print("teacher_forced a2c", {k: out[k].tolist() for k in out if k.endswith("losses")})


def iqn(steps=640, N=4):
    """The reference's IQN + PER (n = 3) run of agent_trace_iqn_per_n3.npz again (asserted: same
    losses), recording for updates {1, 50, 140} what one ``_compute_loss``
    (pfrl/agents/iqn.py:340-400 with :285-338) depends on: the online / target parameters, the
    minibatch, and the three threshold draws it makes (taus, taus_tilde, taus_prime, in the order
    the CPU generator served them) -- and the loss."""
    import tempfile

    import pfrl
    from pfrl import agents, explorers, experiments, replay_buffers
    from pfrl.agents import iqn as riqn

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv

    pfrl.utils.set_random_seed(0)
    env = HostSyntheticAtariVectorEnv(N, seed=13, frame_shape=(12, 12), p_done=0.04)

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    torch.manual_seed(9753)
    q = riqn.ImplicitQuantileQFunction(
        psi=torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU()),
        phi=torch.nn.Sequential(riqn.CosineBasisLinear(16, 32), torch.nn.ReLU()),
        f=torch.nn.Linear(32, 6))
    opt = torch.optim.SGD(q.parameters(), lr=1e-2)
    rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                  num_steps=3, normalize_by_max="memory")
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    ag = agents.IQN(q, opt, rbuf, 0.99, ex, gpu=-1, replay_start_size=40, minibatch_size=8,
                    update_interval=4, target_update_interval=60, phi=phi,
                    batch_accumulator="mean", quantile_thresholds_N=8,
                    quantile_thresholds_N_prime=8, quantile_thresholds_K=4)
    out = {}
    record(ag, out)
    inner = ag._compute_loss
    count = [0]

    def spy(exp_batch, errors_out=None):
        count[0] += 1
        k = count[0]
        if k not in UPDATES:
            return inner(exp_batch, errors_out)
        draws, real = [], torch.rand

        def rec(*a, **kw):
            t = real(*a, **kw)
            draws.append(t.numpy().copy())
            return t

        torch.rand = rec
        try:
            loss = inner(exp_batch, errors_out)
        finally:
            torch.rand = real
        assert len(draws) == 3, len(draws)
        for j, d in enumerate(draws):
            out["u%d_rand%d" % (k, j)] = d
        return loss

    ag._compute_loss = spy
    losses = []
    orig_update = ag.update

    def spy_update(exps, errors_out=None):
        orig_update(exps, errors_out)
        losses.append(float(ag.loss_record[-1]))

    ag.replay_updater.update_func = spy_update
    experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    for k in UPDATES:
        for key in ("state", "next_state"):
            out["u%d_%s" % (k, key)] = _as_u8(out["u%d_%s" % (k, key)])
    finish("iqn_per_n3", out, losses)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    ppo()
    sac()
    td3()
    dqn_like("dqn_uniform_n1", False, 1, False)
    dqn_like("ddqn_per_n3", True, 3, True)
    c51()
    iqn()
    a2c()
    ddpg()
