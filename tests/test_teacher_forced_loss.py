"""Teacher-forced TD-loss parity (VERDICT r3, weak #1a): ONE loss evaluation on the reference's own
pre-update parameters and minibatch, for an early, a middle and a late update of the recorded
runs (tests/golden/make_teacher_forced.py: updates 1, 50, 140 of agent_trace_dqn_uniform_n1 /
_ddqn_per_n3 / _c51_per_n3).  The trajectory tests of test_agent_parity.py can hold late updates
only to a drift tolerance; here every one of them is held to the north-star 1e-5, because nothing
of the trajectory before it enters: reference pfrl/agents/dqn.py:407-470 (``_compute_loss``),
double_dqn.py:11-34, categorical_dqn.py:114-204."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


def _load_flat(module, flat):
    off = 0
    with torch.no_grad():
        for p in module.parameters():
            n = p.numel()
            p.copy_(torch.as_tensor(flat[off:off + n]).view_as(p))
            off += n
    assert off == len(flat)


def _build(kind, gpu):
    from pfrl_amd import agents, explorers, replay_buffers

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    if kind == "c51_per_n3":
        from pfrl_amd.q_functions import DistributionalSingleModelStateQFunctionWithDiscreteAction
        from test_agent_parity import _DistNet

        q = DistributionalSingleModelStateQFunctionWithDiscreteAction(
            _DistNet(), np.linspace(-3, 3, 11, dtype=np.float32))
        opt = torch.optim.SGD(q.parameters(), lr=1e-2)
        rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                      num_steps=3, normalize_by_max="memory")
        return agents.CategoricalDoubleDQN(q, opt, rbuf, 0.99, ex, gpu=gpu, replay_start_size=40,
                                           minibatch_size=8, update_interval=4,
                                           target_update_interval=60, phi=phi,
                                           batch_accumulator="mean")
    from pfrl_amd.q_functions import DiscreteActionValueHead

    q = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU(),
                            torch.nn.Linear(32, 6), DiscreteActionValueHead())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2)
    if kind == "ddqn_per_n3":
        rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                      num_steps=3, normalize_by_max="memory")
        cls = agents.DoubleDQN
    else:
        rbuf = replay_buffers.ReplayBuffer(200, num_steps=1)
        cls = agents.DQN
    return cls(q, opt, rbuf, 0.99, ex, gpu=gpu, replay_start_size=40, minibatch_size=8,
               update_interval=4, target_update_interval=60, phi=phi, batch_accumulator="sum")


def _check(kind, gpu):
    g = np.load(os.path.join(GOLDEN, "teacher_forced_%s.npz" % kind))
    ag = _build(kind, gpu)
    dev = ag.device
    for k in g["updates"]:
        _load_flat(ag.model, g["u%d_params" % k])
        _load_flat(ag.target_model, g["u%d_target_params" % k])
        batch = {}
        for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount",
                    "weights"):
            name = "u%d_%s" % (k, key)
            if name in g.files:
                batch[key] = torch.as_tensor(g[name]).to(dev)
        out = ag._compute_loss(batch, errors_out=None)
        loss = out[0] if isinstance(out, tuple) else out
        want = float(g["u%d_loss" % k])
        got = float(loss.detach().cpu())
        assert abs(got - want) <= TOL * max(1.0, abs(want)), (kind, int(k), got, want)


@pytest.mark.parametrize("kind", ["dqn_uniform_n1", "ddqn_per_n3", "c51_per_n3"])
def test_teacher_forced_losses_on_the_host_path(kind):
    _check(kind, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dqn_uniform_n1", "ddqn_per_n3", "c51_per_n3"])
def test_teacher_forced_losses_on_the_device_path(kind):
    """The fused TD-loss / C51-loss launches (csrc/tdloss.hip, csrc/c51.hip) on the reference's
    parameters and minibatch of updates 1, 50 and 140: each loss within 1e-5."""
    _check(kind, 0)


# ---------------------------------------------------------------------------------------------
# PPO: one evaluation of pfrl/agents/ppo.py:480-532 on the reference's parameters and minibatch
# ---------------------------------------------------------------------------------------------
def _check_ppo(gpu):
    import pfrl_amd as pfrl
    from pfrl_amd import agents
    from pfrl_amd.policies import SoftmaxCategoricalHead

    g = np.load(os.path.join(GOLDEN, "teacher_forced_ppo.npz"))
    clip_eps, vcoef, ecoef = (float(x) for x in g["hyper"])
    torch.manual_seed(4321)
    model = torch.nn.Sequential(
        torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU(),
        pfrl.nn.Branched(torch.nn.Sequential(torch.nn.Linear(32, 6), SoftmaxCategoricalHead()),
                         torch.nn.Linear(32, 1)))
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    ag = agents.PPO(model, opt, gpu=gpu, gamma=0.99, lambd=0.95,
                    phi=lambda x: np.asarray(x, dtype=np.float32) / 255, update_interval=64,
                    minibatch_size=16, epochs=2, clip_eps=clip_eps, clip_eps_vf=None,
                    standardize_advantages=True, max_grad_norm=0.5, value_func_coef=vcoef,
                    entropy_coef=ecoef)
    dev = ag.device
    for k in g["updates"]:
        _load_flat(ag.model, g["u%d_params" % k])
        T = lambda name: torch.as_tensor(g["u%d_%s" % (k, name)]).to(dev)   # noqa: E731
        distribs, vs_pred = ag.model(T("states"))
        records = {}
        loss = ag._lossfun(distribs.entropy(), vs_pred, distribs.log_prob(T("actions")),
                           vs_pred_old=T("vs_pred_old"), log_probs_old=T("log_probs_old"),
                           advs=T("advs"), vs_teacher=T("vs_teacher"), records=records)
        want = g["u%d_losses" % k]
        got = [float(loss.detach().cpu()), float(records["value_loss"].detach().cpu()),
               float(records["policy_loss"].detach().cpu())]
        for a, b, what in zip(got, want, ("loss", "value loss", "policy loss")):
            assert abs(a - b) <= TOL * max(1.0, abs(b)), (int(k), what, a, float(b))


def test_teacher_forced_ppo_losses_on_the_host_path():
    """the trajectory trace holds PPO's losses to 1e-5 only for the
    first rollout; here minibatch updates 1, 9, 20 and 32 of that same reference run are each held
    to 1e-5 on the reference's own parameters, states, actions and dataset columns."""
    _check_ppo(-1)


@pytest.mark.gpu
def test_teacher_forced_ppo_losses_on_the_device_path():
    _check_ppo(0)


# ---------------------------------------------------------------------------------------------
# SAC: one update (pfrl/agents/soft_actor_critic.py:214-300) on the reference's five networks and
# minibatch; sampling noise off on both sides, as in the trajectory trace
# ---------------------------------------------------------------------------------------------
def _check_sac(gpu, **agent_kw):
    from test_agent_parity import _NoNoise, _squashed_head

    from pfrl_amd import agents, replay_buffers
    from pfrl_amd.nn import ConcatObsAndAction, Lambda

    g = np.load(os.path.join(GOLDEN, "teacher_forced_sac.npz"))
    obs_dim, act_dim = 24, 3
    torch.manual_seed(1357)
    policy = torch.nn.Sequential(torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(),
                                 torch.nn.Linear(32, act_dim * 2), Lambda(_squashed_head))

    def q():
        return torch.nn.Sequential(ConcatObsAndAction(), torch.nn.Linear(obs_dim + act_dim, 32),
                                   torch.nn.ReLU(), torch.nn.Linear(32, 1))

    q1, q2 = q(), q()
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
    ag = agents.SoftActorCritic(
        policy, q1, q2, opts[0], opts[1], opts[2], replay_buffers.ReplayBuffer(500), gamma=0.99,
        gpu=gpu, replay_start_size=40, minibatch_size=16, update_interval=1,
        burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
        entropy_target=None, initial_temperature=0.2, soft_update_tau=5e-3, **agent_kw)
    dev = ag.device
    seen = {}
    orig_record = ag._record_stats

    def spy_record(st):
        orig_record(st)
        for name in ("loss1", "loss2", "policy_loss"):
            if name in st:
                seen[name] = float(st[name].detach().reshape(-1)[0].cpu())

    ag._record_stats = spy_record
    for k in g["updates"]:
        for name, m in (("policy", ag.policy), ("q1", ag.q_func1), ("q2", ag.q_func2),
                        ("tq1", ag.target_q_func1), ("tq2", ag.target_q_func2)):
            _load_flat(m, g["u%d_%s_params" % (k, name)])
        batch = {key: torch.as_tensor(g["u%d_%s" % (k, key)]).to(dev)
                 for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount")}
        seen.clear()
        with _NoNoise():
            ag._update_impl(batch)
        want = list(g["u%d_q_losses" % k]) + [float(g["u%d_policy_loss" % k])]
        got = [seen["loss1"], seen["loss2"], seen["policy_loss"]]
        for a, b, what in zip(got, want, ("Q1 loss", "Q2 loss", "policy loss")):
            assert abs(a - b) <= TOL * max(1.0, abs(b)), (int(k), what, a, float(b))


def test_teacher_forced_sac_losses_on_the_host_path():
    """updates 1, 90 and 180 of the reference's SAC run, each on the reference's
    own parameters (policy, twin Q, twin target Q) and minibatch: both critic losses and the policy
    loss within 1e-5 (the trajectory trace holds the critic losses to 1e-4 and never looked at the
    policy loss)."""
    _check_sac(-1)


@pytest.mark.gpu
def test_teacher_forced_sac_losses_on_the_device_path():
    """... through the twin-Q MFMA launches, the fused squashed-Gaussian head and the loss kernels
    of csrc/actor.hip."""
    _check_sac(0)


def _check_td3(gpu, **agent_kw):
    from test_agent_parity import _shifted_smoothing

    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers

    g = np.load(os.path.join(GOLDEN, "teacher_forced_td3.npz"))
    obs_dim, act_dim = 24, 3
    torch.manual_seed(2468)
    policy = torch.nn.Sequential(
        torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(), torch.nn.Linear(32, act_dim),
        pfrl.nn.BoundByTanh(low=-np.ones(act_dim, dtype=np.float32),
                            high=np.ones(act_dim, dtype=np.float32)),
        pfrl.policies.DeterministicHead())

    def q():
        return torch.nn.Sequential(pfrl.nn.ConcatObsAndAction(),
                                   torch.nn.Linear(obs_dim + act_dim, 32), torch.nn.ReLU(),
                                   torch.nn.Linear(32, 1))

    q1, q2 = q(), q()
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
    ag = agents.TD3(policy, q1, q2, opts[0], opts[1], opts[2], replay_buffers.ReplayBuffer(500),
                    gamma=0.99, explorer=explorers.AdditiveGaussian(scale=0.1, low=-1.0, high=1.0),
                    gpu=gpu, replay_start_size=40, minibatch_size=16, update_interval=1,
                    soft_update_tau=5e-3,
                    burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
                    policy_update_delay=2, target_policy_smoothing_func=_shifted_smoothing, **agent_kw)
    dev = ag.device
    seen = {}
    orig_record = ag._record_stats

    def spy_record(st):
        orig_record(st)
        for name in ("loss1", "loss2", "policy_loss"):
            if name in st:
                seen[name] = float(st[name].detach().reshape(-1)[0].cpu())

    ag._record_stats = spy_record
    n_policy = 0
    for k in g["updates"]:
        for name, m in (("policy", ag.policy), ("q1", ag.q_func1), ("q2", ag.q_func2),
                        ("tpolicy", ag.target_policy), ("tq1", ag.target_q_func1),
                        ("tq2", ag.target_q_func2)):
            _load_flat(m, g["u%d_%s_params" % (k, name)])
        batch = {key: torch.as_tensor(g["u%d_%s" % (k, key)]).to(dev)
                 for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount")}
        with_policy = bool(int(g["u%d_with_policy" % k]))
        seen.clear()
        ag._update_impl(batch, with_policy)
        want = list(g["u%d_q_losses" % k])
        got = [seen["loss1"], seen["loss2"]]
        names = ["Q1 loss", "Q2 loss"]
        if with_policy:
            n_policy += 1
            want.append(float(g["u%d_policy_loss" % k]))
            got.append(seen["policy_loss"])
            names.append("policy loss")
        else:
            assert "policy_loss" not in seen
        for a, b, what in zip(got, want, names):
            assert abs(a - b) <= TOL * max(1.0, abs(b)), (int(k), what, a, float(b))
    assert n_policy >= 2


def test_teacher_forced_td3_losses_on_the_host_path():
    """updates 1, 90 and 180 of the reference's TD3 run, each on the
    reference's own six networks (policy, twin Q, their targets) and minibatch: both critic losses
    and -- on the delayed policy updates -- the policy loss within 1e-5 (the trajectory trace holds
    the critic losses to 1e-4 over the run and never looked at the policy loss)."""
    _check_td3(-1)


@pytest.mark.gpu
def test_teacher_forced_td3_losses_on_the_device_path():
    """... through the twin-Q launches and the MFMA linear kernels on the device."""
    _check_td3(0)


# ---------------------------------------------------------------------------------------------
# IQN: one evaluation of pfrl/agents/iqn.py:340-400 (with :285-338) on the reference's parameters,
# minibatch AND threshold draws (the three torch.rand calls of one _compute_loss, in order)
# ---------------------------------------------------------------------------------------------
def _check_iqn(gpu, **agent_kw):
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.agents import iqn

    g = np.load(os.path.join(GOLDEN, "teacher_forced_iqn_per_n3.npz"))

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    q = iqn.ImplicitQuantileQFunction(
        psi=torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU()),
        phi=torch.nn.Sequential(iqn.CosineBasisLinear(16, 32), torch.nn.ReLU()),
        f=torch.nn.Linear(32, 6))
    opt = torch.optim.SGD(q.parameters(), lr=1e-2)
    rbuf = replay_buffers.PrioritizedReplayBuffer(200, alpha=0.5, beta0=0.4, betasteps=100,
                                                  num_steps=3, normalize_by_max="memory")
    ex = explorers.LinearDecayEpsilonGreedy(1.0, 0.1, 400, lambda: np.random.randint(6))
    ag = agents.IQN(q, opt, rbuf, 0.99, ex, gpu=gpu, replay_start_size=40, minibatch_size=8,
                    update_interval=4, target_update_interval=60, phi=phi,
                    batch_accumulator="mean", quantile_thresholds_N=8,
                    quantile_thresholds_N_prime=8, quantile_thresholds_K=4, **agent_kw)
    dev = ag.device
    for k in g["updates"]:
        _load_flat(ag.model, g["u%d_params" % k])
        _load_flat(ag.target_model, g["u%d_target_params" % k])
        batch = {}
        for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount",
                    "weights"):
            a = g["u%d_%s" % (k, key)]
            if key in ("state", "next_state"):          # stored as the u8 frames: phi gives them back
                a = a.astype(np.float32) / 255
            batch[key] = torch.as_tensor(a).to(dev)
        draws = [torch.as_tensor(g["u%d_rand%d" % (k, j)]).to(dev) for j in range(3)]

        def replay(rows, cols, _d=draws):
            t = _d.pop(0)
            assert tuple(t.shape) == (rows, cols), (tuple(t.shape), rows, cols)
            return t

        ag._rand = replay
        loss = ag._compute_loss(batch, errors_out=None)[0]
        assert not draws
        want, got = float(g["u%d_loss" % k]), float(loss.detach().cpu())
        assert abs(got - want) <= TOL * max(1.0, abs(want)), (int(k), got, want)


def test_teacher_forced_iqn_losses_on_the_host_path():
    _check_iqn(-1)


@pytest.mark.gpu
def test_teacher_forced_iqn_losses_on_the_device_path():
    """The quantile-regression loss on the device (thresholds replayed from the reference's CPU
    generator) at updates 1, 50 and 140 of the recorded run: each within 1e-5."""
    _check_iqn(0, use_graphs=False)


# ---------------------------------------------------------------------------------------------
# A2C: one whole update of pfrl/agents/a2c.py:169-213 on the reference's parameters and rollout
# storage: the three loss terms, the returns, and the parameters after the clipped SGD step
# ---------------------------------------------------------------------------------------------
def _check_a2c(gpu):
    import pfrl_amd as pfrl
    from pfrl_amd import agents
    from pfrl_amd.policies import SoftmaxCategoricalHead

    g = np.load(os.path.join(GOLDEN, "teacher_forced_a2c.npz"))

    def phi(x):
        return np.asarray(x, dtype=np.float32) / 255

    T, N = 5, 4
    for use_gae in (True, False):
        tag = "gae%d" % int(use_gae)
        for k in g[tag + "_updates"]:
            pre = "%s_u%d_" % (tag, k)
            model = torch.nn.Sequential(
                torch.nn.Flatten(), torch.nn.Linear(4 * 144, 32), torch.nn.ReLU(),
                pfrl.nn.Branched(torch.nn.Sequential(torch.nn.Linear(32, 6), SoftmaxCategoricalHead()),
                                 torch.nn.Linear(32, 1)))
            opt = torch.optim.SGD(model.parameters(), lr=1e-2)
            ag = agents.A2C(model, opt, gamma=0.99, num_processes=N, gpu=gpu, update_steps=T, phi=phi,
                            use_gae=use_gae, tau=0.95, max_grad_norm=0.5)
            dev = ag.device
            _load_flat(ag.model, g[pre + "params"])
            states = torch.as_tensor(g[pre + "states"].astype(np.float32) / 255).to(dev)
            actions = torch.as_tensor(g[pre + "actions"]).to(dev)
            if gpu >= 0:
                # the device path stores frame-slot refs and gathers the network input from HBM
                # (tested on its own); here the "slots" index the reference's recorded states
                table = states.reshape((T + 1) * N, *states.shape[2:])
                ag._flush_storage(N, 1, actions[0])
                ag.refs = torch.arange((T + 1) * N, dtype=torch.int32, device=dev).view(T + 1, N, 1)
                ag._gather = lambda refs, _t=table: _t[refs.reshape(-1).long()]
            else:
                ag._flush_storage(N, tuple(states.shape[2:]), actions[0])
                ag.refs = states.clone()
            ag.actions = actions.reshape(ag.actions.shape).float().clone()
            ag.rewards = torch.as_tensor(g[pre + "rewards"]).to(dev).clone()
            ag.masks = torch.as_tensor(g[pre + "masks"]).to(dev).clone()
            ag.value_preds = torch.as_tensor(g[pre + "value_preds"]).to(dev).clone()
            ag.update()
            got = np.asarray([float(v.cpu()) for v in ag._last_losses])
            np.testing.assert_allclose(got, g[pre + "losses"], rtol=TOL, atol=TOL, err_msg=pre)
            np.testing.assert_allclose(ag.returns.cpu().numpy(), g[pre + "returns"], rtol=TOL, atol=TOL)
            after = np.concatenate([p.detach().cpu().numpy().ravel() for p in ag.model.parameters()])
            np.testing.assert_allclose(after, g[pre + "params_after"], rtol=TOL, atol=1e-6, err_msg=pre)


def test_teacher_forced_a2c_update_on_the_host_path():
    _check_a2c(-1)


@pytest.mark.gpu
def test_teacher_forced_a2c_update_on_the_device_path():
    """Return scan (pfrl_a2c_returns), losses and the clipped step on the device, on the
    reference's storage of updates 1, 12, 30 (GAE) and 1, 30 (n-step returns): 1e-5."""
    _check_a2c(0)


# ---------------------------------------------------------------------------------------------
# DDPG: one update of pfrl/agents/ddpg.py:150-200 on the reference's four networks and minibatch
# ---------------------------------------------------------------------------------------------
def _check_ddpg(gpu, **agent_kw):
    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers

    g = np.load(os.path.join(GOLDEN, "teacher_forced_ddpg.npz"))
    obs_dim, act_dim = 24, 3
    torch.manual_seed(2468)
    policy = torch.nn.Sequential(
        torch.nn.Linear(obs_dim, 32), torch.nn.ReLU(), torch.nn.Linear(32, act_dim),
        pfrl.nn.BoundByTanh(low=-np.ones(act_dim, dtype=np.float32),
                            high=np.ones(act_dim, dtype=np.float32)),
        pfrl.policies.DeterministicHead())
    q1 = torch.nn.Sequential(pfrl.nn.ConcatObsAndAction(), torch.nn.Linear(obs_dim + act_dim, 32),
                             torch.nn.ReLU(), torch.nn.Linear(32, 1))
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1)]
    ag = agents.DDPG(policy, q1, opts[0], opts[1], replay_buffers.ReplayBuffer(500), gamma=0.99,
                     explorer=explorers.AdditiveGaussian(scale=0.1, low=-1.0, high=1.0), gpu=gpu,
                     replay_start_size=40, minibatch_size=16, update_interval=1,
                     target_update_interval=7, target_update_method="soft", soft_update_tau=5e-2,
                     burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
                     **agent_kw)
    dev = ag.device
    seen = {}
    orig_record = ag._record_stats

    def spy_record(st):
        orig_record(st)
        for name in ("critic_loss", "actor_loss"):
            if name in st:
                seen[name] = float(st[name].detach().reshape(-1)[0].cpu())

    ag._record_stats = spy_record
    for k in g["updates"]:
        for name, m in (("policy", ag.policy), ("q", ag.q_function), ("tpolicy", ag.target_policy),
                        ("tq", ag.target_q_function)):
            _load_flat(m, g["u%d_%s_params" % (k, name)])
        batch = {key: torch.as_tensor(g["u%d_%s" % (k, key)]).to(dev)
                 for key in ("state", "action", "reward", "next_state", "is_state_terminal", "discount")}
        seen.clear()
        ag._update_impl(batch)
        for name in ("critic_loss", "actor_loss"):
            want = float(g["u%d_%s" % (k, name)])
            assert abs(seen[name] - want) <= TOL * max(1.0, abs(want)), (int(k), name, seen[name], want)


def test_teacher_forced_ddpg_losses_on_the_host_path():
    """Updates 1, 90 and 180 of the reference's DDPG run on its own networks (policy, Q, their
    targets) and minibatch: the critic loss and the actor loss (after the critic's step) at 1e-5."""
    _check_ddpg(-1)


@pytest.mark.gpu
def test_teacher_forced_ddpg_losses_on_the_device_path():
    _check_ddpg(0)
