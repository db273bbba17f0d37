"""Host-side pieces of the API mirror that need no GPU: small modules, explorers,
distributions, action values, the gym-free CartPole (known answers)."""
import math
import os

import numpy as np
import pytest
import torch

import pfrl_amd as pfrl


def test_delta_distribution_and_deterministic_head():
    """pfrl/distributions/delta.py: sampling returns loc (rsample differentiably),
    densities are undefined; DeterministicHead wraps it in Independent(…, 1)."""
    from pfrl_amd.distributions import Delta

    loc = torch.tensor([[0.5, -1.0], [2.0, 3.0]], requires_grad=True)
    d = Delta(loc)
    assert torch.equal(d.mean, loc) and torch.equal(d.stddev, torch.zeros_like(loc))
    assert d.sample().requires_grad is False and torch.equal(d.sample(), loc.detach())
    (d.rsample() * torch.tensor([[1.0, 2.0], [3.0, 4.0]])).sum().backward()
    assert torch.equal(loc.grad, torch.tensor([[1.0, 2.0], [3.0, 4.0]]))
    assert d.rsample((3,)).shape == (3, 2, 2)
    assert d.expand((4, 2, 2)).loc.shape == (4, 2, 2)
    for fn in (lambda: d.log_prob(loc), d.entropy):
        with pytest.raises(RuntimeError):
            fn()
    head = pfrl.policies.DeterministicHead()(loc.detach())
    assert isinstance(head, torch.distributions.Independent) and head.event_shape == (2,)
    assert torch.equal(head.sample(), loc.detach())


def test_bound_by_tanh():
    low, high = np.array([-2.0, 0.0], dtype=np.float32), np.array([2.0, 10.0], dtype=np.float32)
    x = torch.tensor([[0.0, 0.0], [100.0, -100.0], [0.3, 1.2]])
    y = pfrl.nn.BoundByTanh(low, high)(x)
    np.testing.assert_allclose(y[0].numpy(), [0.0, 5.0], atol=1e-6)
    np.testing.assert_allclose(y[1].numpy(), [2.0, 0.0], atol=1e-5)
    np.testing.assert_allclose(y[2].numpy(), [2 * math.tanh(0.3), 5 * math.tanh(1.2) + 5],
                               rtol=1e-6)
    assert torch.equal(y, pfrl.nn.bound_by_tanh.bound_by_tanh(x, low, high))
    assert torch.equal(y, pfrl.functions.bound_by_tanh.bound_by_tanh(x, low, high))


def test_additive_gaussian_explorer_uses_the_numpy_stream():
    """pfrl/explorers/additive_gaussian.py:27-34: one np.random.normal draw of the action's
    shape, float32, clipped to [low, high] when given."""
    ex = pfrl.explorers.AdditiveGaussian(scale=0.5, low=-1.0, high=1.0)
    a = np.array([0.9, -0.2, 0.0], dtype=np.float32)
    np.random.seed(3)
    got = ex.select_action(0, lambda: a)
    np.random.seed(3)
    want = np.clip(a + np.random.normal(scale=0.5, size=3).astype(np.float32), -1.0, 1.0)
    np.testing.assert_array_equal(got, want)
    free = pfrl.explorers.AdditiveGaussian(scale=0.5)
    np.random.seed(3)
    np.testing.assert_array_equal(free.select_action(0, lambda: a),
                                  a + np.random.RandomState(3).normal(scale=0.5, size=3).astype(
                                      np.float32))
    assert "AdditiveGaussian" in repr(ex)


def test_quantile_discrete_action_value():
    """pfrl/action_value.py:183-229."""
    from pfrl_amd.action_value import QuantileDiscreteActionValue

    q = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)   # (batch, taus, actions)
    av = QuantileDiscreteActionValue(q)
    assert torch.equal(av.q_values, q.mean(1))
    assert av.greedy_actions.tolist() == [3, 3] and av.n_actions == 4
    z = av.evaluate_actions_as_quantiles(torch.tensor([1, 2]))
    assert torch.equal(z, torch.stack([q[0, :, 1], q[1, :, 2]]))
    assert torch.equal(av.max, q.mean(1)[:, 3])
    assert av[1:].quantiles.shape == (1, 3, 4)
    assert av.params == (q,)


def test_iqn_building_blocks():
    """cosine embedding (i = 1..n), the quantile Huber loss and its accumulation
    (pfrl/agents/iqn.py:11-60, 176-255) on hand-computable inputs."""
    from pfrl_amd.agents import iqn

    x = torch.tensor([[0.0, 0.5]])
    emb = iqn.cosine_basis_functions(x, 3)
    np.testing.assert_allclose(emb[0, 0].numpy(), [1.0, 1.0, 1.0], atol=1e-6)
    np.testing.assert_allclose(emb[0, 1].numpy(), [0.0, -1.0, 0.0], atol=1e-6)
    y = torch.tensor([[0.0, 2.0]])            # (B=1, N=2)
    t = torch.tensor([[0.5, 3.0, -4.0]])      # (B=1, N'=3)
    taus = torch.tensor([[0.25, 0.75]])
    el = iqn.compute_eltwise_huber_quantile_loss(y, t, taus)
    assert el.shape == (1, 2, 3)

    def huber(d):
        return 0.5 * d * d if abs(d) < 1 else abs(d) - 0.5

    want = np.zeros((2, 3))
    for i, (yy, tau) in enumerate(zip([0.0, 2.0], [0.25, 0.75])):
        for j, tt in enumerate([0.5, 3.0, -4.0]):
            want[i, j] = abs(tau - (1.0 if tt < yy else 0.0)) * huber(yy - tt)
    np.testing.assert_allclose(el[0].numpy(), want, rtol=1e-6)
    np.testing.assert_allclose(iqn.compute_value_loss(el, "sum").item(), want.mean(1).sum(),
                               rtol=1e-6)
    np.testing.assert_allclose(iqn.compute_value_loss(el, "mean").item(), want.mean(1).sum(),
                               rtol=1e-6)   # B = 1
    w = torch.tensor([0.5])
    np.testing.assert_allclose(iqn.compute_weighted_value_loss(el, w, "mean").item(),
                               0.5 * want.mean(1).sum(), rtol=1e-6)
    lin = iqn.CosineBasisLinear(4, 5)
    assert lin(torch.rand(2, 3)).shape == (2, 3, 5)


def test_cartpole_known_step_and_limits():
    """CartPole-v1 constants: one Euler step from a known state (values computed by hand
    from the published equations), termination thresholds and the 500-step truncation."""
    from pfrl_amd.envs import CartPoleEnv

    env = CartPoleEnv(seed=0)
    obs = env.reset()
    assert obs.dtype == np.float32 and obs.shape == (4,) and np.all(np.abs(obs) <= 0.05)
    env.state = np.array([0.0, 0.0, 0.1, 0.0])
    o, r, done, info = env.step(1)
    s, c = math.sin(0.1), math.cos(0.1)
    tmp = (10.0 + 0.05 * 0.0 * s) / 1.1
    th_acc = (9.8 * s - c * tmp) / (0.5 * (4.0 / 3.0 - 0.1 * c * c / 1.1))
    x_acc = tmp - 0.05 * th_acc * c / 1.1
    np.testing.assert_allclose(o, [0.0, 0.02 * x_acc, 0.1, 0.02 * th_acc], rtol=1e-6, atol=1e-9)
    assert r == 1.0 and done is False and info == {}
    env.state = np.array([2.39, 5.0, 0.0, 0.0])
    assert env.step(1)[2] is True                       # |x| > 2.4
    env.reset()
    env.state = np.array([0.0, 0.0, 0.2085, 1.0])
    assert env.step(0)[2] is True                       # |theta| > 12 degrees
    env = CartPoleEnv(seed=1, max_episode_steps=3)
    env.reset()
    infos = []
    for _ in range(3):
        env.state = np.zeros(4)
        infos.append(env.step(0)[3])
    assert infos[:2] == [{}, {}] and infos[2] == {"needs_reset": True}
    assert env.action_space.n == 2 and env.action_space.sample() in (0, 1)


def test_fuse_conv_bias_relu_keeps_parameters_and_outputs_on_cpu():
    nn = torch.nn
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(4, 8, 3), nn.ReLU(), nn.Conv2d(8, 8, 3), nn.Tanh(),
                      nn.Conv2d(8, 4, 3), nn.ReLU(), nn.Flatten(), nn.Linear(4 * 4, 3))
    x = torch.rand(2, 4, 8, 8)
    y = m(x)
    keys = list(m.state_dict().keys())
    ids = [id(p) for p in m.parameters()]
    f = pfrl.nn.fuse_conv_bias_relu(m)
    assert list(f.state_dict().keys()) == keys and [id(p) for p in f.parameters()] == ids
    kinds = [type(c).__name__ for c in f]
    assert kinds[0] == "_ConvSlot" and kinds[1] == "_Identity"       # Conv2d + ReLU fused
    assert kinds[2] == "Conv2d" and kinds[3] == "Tanh"               # not a ReLU: untouched
    assert kinds[4] == "_ConvSlot"
    assert torch.equal(f(x), y)
    from pfrl_amd.nn.atari_cnn import wants_channels_last

    assert not wants_channels_last(f)
    assert wants_channels_last(f.to(memory_format=torch.channels_last))
    assert not wants_channels_last(nn.Linear(3, 3))


def test_soft_target_sync_matches_per_tensor_formula():
    """pfrl/utils/copy_param.py:9-24: theta' <- (1 - tau) theta' + tau theta."""
    from pfrl_amd.utils.copy_param import synchronize_parameters

    torch.manual_seed(1)
    src, dst = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    want = {k: (1 - 0.1) * v + 0.1 * src.state_dict()[k] for k, v in dst.state_dict().items()}
    synchronize_parameters(src=src, dst=dst, method="soft", tau=0.1)
    for k, v in dst.state_dict().items():
        np.testing.assert_allclose(v.numpy(), want[k].numpy(), rtol=1e-6)
    synchronize_parameters(src=src, dst=dst, method="hard")
    for k, v in dst.state_dict().items():
        assert torch.equal(v, src.state_dict()[k])
    with pytest.raises(ValueError):
        synchronize_parameters(src=src, dst=dst, method="nope")


def test_boltzmann_and_ou_explorers():
    """pfrl/explorers/boltzmann.py:19-27, additive_ou.py:35-59: NumPy-stream draws."""
    from pfrl_amd.action_value import DiscreteActionValue

    av = DiscreteActionValue(torch.tensor([[1.0, 2.0, 4.0]]))
    ex = pfrl.explorers.Boltzmann(T=2.0)
    np.random.seed(5)
    got = [ex.select_action(0, None, action_value=av) for _ in range(5)]
    np.random.seed(5)
    p = torch.softmax(av.q_values / 2.0, dim=-1).numpy().ravel()
    assert got == [np.random.choice(np.arange(3), p=p) for _ in range(5)]
    with pytest.raises(AssertionError):
        ex.select_action(0, None)
    ou = pfrl.explorers.AdditiveOU(mu=0.5, theta=0.2, sigma=0.1, start_with_mu=True)
    a = np.zeros(2, dtype=np.float32)
    np.testing.assert_array_equal(ou.select_action(0, lambda: a), [0.5, 0.5])
    np.random.seed(1)
    second = ou.select_action(1, lambda: a)
    np.random.seed(1)
    want = 0.5 + 0.2 * (0.5 - 0.5) + np.random.normal(size=2, loc=0, scale=0.1)
    np.testing.assert_allclose(second, want, rtol=1e-6)
    ou2 = pfrl.explorers.AdditiveOU(theta=0.15, sigma=0.3)
    np.random.seed(2)
    first = ou2.select_action(0, lambda: a)
    np.random.seed(2)
    np.testing.assert_allclose(first, np.random.normal(size=2, loc=0.0, scale=0.3 / np.sqrt(
        2 * 0.15 - 0.15 ** 2)).astype(np.float32), rtol=1e-6)


def test_gaussian_head_with_fixed_covariance_and_hooks():
    d = pfrl.policies.GaussianHeadWithFixedCovariance(scale=0.5)(torch.zeros(3, 2))
    assert d.event_shape == (2,) and torch.allclose(d.stddev, torch.full((3, 2), 0.5))
    np.testing.assert_allclose(d.log_prob(torch.zeros(3, 2)).numpy(),
                               2 * (-math.log(0.5) - 0.5 * math.log(2 * math.pi)), rtol=1e-6)
    from pfrl_amd import experiments

    with pytest.raises(TypeError):
        experiments.StepHook()
    with pytest.raises(TypeError):
        experiments.EvaluationHook()
    assert issubclass(type(experiments.LinearInterpolationHook(10, 1.0, 0.0, lambda *a: None)),
                      object)
    seen = []
    hook = experiments.LinearInterpolationHook(11, 1.0, 0.0, lambda env, agent, v: seen.append(v))
    for step in (1, 6, 11):
        hook(None, None, step)
    np.testing.assert_allclose(seen, [1.0, 0.5, 0.0])


class _ToyEnv:
    """obs = float64 step counter, reward 2, never done."""

    class _Space:
        n = 3
        low = np.array([-2.0, 0.0])
        high = np.array([2.0, 10.0])

    action_space = _Space()

    def __init__(self):
        self.t = 0
        self.actions = []

    def reset(self):
        self.t = 0
        return np.array([0.0])

    def step(self, action):
        self.t += 1
        self.actions.append(action)
        return np.array([float(self.t)]), 2.0, False, {}


def test_gym_free_env_wrappers():
    """Behaviour of pfrl/wrappers/{continuing_time_limit,cast_observation,scale_reward,
    randomize_action,normalize_action_space}.py on a toy env."""
    from pfrl_amd import wrappers

    env = wrappers.ContinuingTimeLimit(_ToyEnv(), max_episode_steps=3)
    with pytest.raises(AssertionError):
        env.step(0)
    env.reset()
    infos = [env.step(0)[3] for _ in range(4)]
    assert infos[:2] == [{}, {}] and infos[2] == {"needs_reset": True} == infos[3]
    assert env.step(0)[2] is False and env.t == 5                 # attribute delegation
    env.reset()
    assert env.step(0)[3] == {}

    cast = wrappers.CastObservationToFloat32(_ToyEnv())
    assert cast.reset().dtype == np.float32
    o = cast.step(0)[0]
    assert o.dtype == np.float32 and cast.original_observation.dtype == np.float64

    sc = wrappers.ScaleReward(_ToyEnv(), 0.25)
    sc.reset()
    assert sc.step(0)[1] == 0.5 and sc.original_reward == 2.0

    base = _ToyEnv()
    ra = wrappers.RandomizeAction(base, 0.5)
    ra.seed(7)
    ra.reset()
    for _ in range(20):
        ra.step(1)
    rs = np.random.RandomState(7)
    want = [rs.randint(3) if rs.rand() < 0.5 else 1 for _ in range(20)]
    assert base.actions == want
    with pytest.raises(AssertionError):
        wrappers.RandomizeAction(_ToyEnv(), 1.5)

    base = _ToyEnv()
    na = wrappers.NormalizeActionSpace(base)
    np.testing.assert_array_equal(na.action_space.low, [-1.0, -1.0])
    na.reset()
    na.step(np.array([-1.0, 1.0]))
    na.step(np.array([0.0, 0.0]))
    np.testing.assert_allclose(base.actions[0], [-2.0, 10.0])
    np.testing.assert_allclose(base.actions[1], [0.0, 5.0])


def test_distributional_fc_q_function():
    q = pfrl.q_functions.DistributionalFCStateQFunctionWithDiscreteAction(
        5, 3, 11, -2.0, 2.0, n_hidden_channels=8, n_hidden_layers=2)
    av = q(torch.rand(4, 5))
    assert av.q_dist.shape == (4, 3, 11)
    np.testing.assert_allclose(av.q_dist.sum(dim=2).detach().numpy(), 1.0, rtol=1e-5)
    np.testing.assert_allclose(av.z_values.numpy(), np.linspace(-2, 2, 11), rtol=1e-6)
    assert av.q_values.shape == (4, 3)
    import pickle

    pickle.loads(pickle.dumps(q))


def test_empirical_normalization_matches_batch_statistics():
    """pfrl/nn/empirical_normalization.py: after seeing batches the running mean / variance
    equal those of all data seen; clip and ``until``; forward(update=False) leaves them."""
    torch.manual_seed(0)
    en = pfrl.nn.EmpiricalNormalization(3, clip_threshold=2.0)
    xs = [torch.randn(n, 3) * 2 + 1 for n in (5, 17, 1)]
    for x in xs:
        en.experience(x)
    allx = torch.cat(xs)
    np.testing.assert_allclose(en.mean.numpy(), allx.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(en.std.numpy(), allx.std(0, unbiased=False).numpy(), rtol=1e-5)
    assert int(en.count) == 23
    y = en(allx, update=False)
    assert int(en.count) == 23 and float(y.abs().max()) <= 2.0
    want = torch.clamp((allx - allx.mean(0)) / torch.sqrt(allx.var(0, unbiased=False) + 1e-2),
                       -2.0, 2.0)
    np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(en.inverse((allx - en.mean) / torch.sqrt(en.std ** 2 + 1e-2)).numpy(),
                               allx.numpy(), rtol=1e-4, atol=1e-4)
    lim = pfrl.nn.EmpiricalNormalization(3, until=10)
    lim(torch.randn(8, 3))
    lim(torch.randn(8, 3))
    m = lim.mean.clone()
    lim(torch.randn(8, 3))            # count 16 >= 10: frozen
    assert int(lim.count) == 16 and torch.equal(lim.mean, m)
    assert set(en.state_dict().keys()) == {"_mean", "_var", "count"}


def _tiny_continuous_agents():
    from pfrl_amd import agents, explorers, replay_buffers

    obs_dim, act_dim = 5, 2

    def policy_det():
        return torch.nn.Sequential(
            torch.nn.Linear(obs_dim, 8), torch.nn.ReLU(), torch.nn.Linear(8, act_dim),
            pfrl.nn.BoundByTanh(low=-np.ones(act_dim, dtype=np.float32),
                                high=np.ones(act_dim, dtype=np.float32)),
            pfrl.policies.DeterministicHead())

    def qf():
        return torch.nn.Sequential(pfrl.nn.ConcatObsAndAction(),
                                   torch.nn.Linear(obs_dim + act_dim, 8), torch.nn.ReLU(),
                                   torch.nn.Linear(8, 1))

    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=1e-2)
    ex = explorers.AdditiveGaussian(0.1, -1.0, 1.0)
    p, q1, q2 = policy_det(), qf(), qf()
    td3 = agents.TD3(p, q1, q2, sgd(p), sgd(q1), sgd(q2), replay_buffers.ReplayBuffer(100),
                     gamma=0.99, explorer=ex, gpu=-1, replay_start_size=10, minibatch_size=4)
    p, q = policy_det(), qf()
    ddpg = agents.DDPG(p, q, sgd(p), sgd(q), replay_buffers.ReplayBuffer(100), gamma=0.99,
                       explorer=ex, gpu=-1, replay_start_size=10, minibatch_size=4)
    gp = torch.nn.Sequential(
        torch.nn.Linear(obs_dim, 8), torch.nn.ReLU(), torch.nn.Linear(8, act_dim),
        pfrl.policies.GaussianHeadWithFixedCovariance(0.3))
    q1, q2 = qf(), qf()
    sac = agents.SoftActorCritic(gp, q1, q2, sgd(gp), sgd(q1), sgd(q2),
                                 replay_buffers.ReplayBuffer(100), gamma=0.99, gpu=-1,
                                 replay_start_size=10, minibatch_size=4, entropy_target=-2.0,
                                 temperature_optimizer_lr=1e-3)
    return obs_dim, act_dim, dict(td3=td3, ddpg=ddpg, sac=sac)


def test_continuous_agents_train_save_load_on_the_host(tmp_path):
    """TD3 / DDPG / SAC in host mode: a few dozen updates, statistics names of the
    reference, save() / load() round trip of every saved attribute."""
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv

    obs_dim, act_dim, ags = _tiny_continuous_agents()
    want_stats = {
        "td3": ["average_q1", "average_q2", "average_q_func1_loss", "average_q_func2_loss",
                "average_policy_loss", "policy_n_updates", "q_func_n_updates"],
        "ddpg": ["average_q", "average_actor_loss", "average_critic_loss", "n_updates"],
        "sac": ["average_q1", "average_q2", "average_q_func1_loss", "average_q_func2_loss",
                "n_updates", "average_entropy", "temperature"],
    }
    for name, ag in ags.items():
        env = HostSyntheticVectorObsEnv(2, obs_dim=obs_dim, act_dim=act_dim, seed=1, p_done=0.1)
        pfrl.experiments.train_agent_batch(ag, env, 60, str(tmp_path / ("run_" + name)))
        stats = ag.get_statistics()
        assert [k for k, _ in stats] == want_stats[name]
        assert all(np.isfinite(float(v)) for _, v in stats), (name, stats)
        d = str(tmp_path / ("save_" + name))
        ag.save(d)
        before = {k: v.clone() for k, v in ag._policy().state_dict().items()}
        with torch.no_grad():
            for p in ag._policy().parameters():
                p.add_(1.0)
        ag.load(d)
        for k, v in ag._policy().state_dict().items():
            assert torch.equal(v, before[k]), (name, k)
        with ag.eval_mode():
            a = ag.batch_act([np.zeros(obs_dim, dtype=np.float32)] * 3)
            assert np.asarray(a).shape == (3, act_dim)
            ag.batch_observe([np.zeros(obs_dim, dtype=np.float32)] * 3, [0.0] * 3, [False] * 3,
                             [False] * 3)


def test_rmsprop_eps_inside_sqrt_matches_reference_steps():
    """tests/golden/rmsprop_eps_inside_sqrt.npz: parameters after each of four steps and the final
    optimizer state, for the plain / centered / momentum / weight-decay configurations."""
    from pfrl_amd import optimizers

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rmsprop_eps_inside_sqrt.npz"))
    configs = dict(plain=dict(), centered=dict(centered=True), momentum=dict(momentum=0.9),
                   decay=dict(weight_decay=0.01, centered=True, momentum=0.5))
    for name, kw in configs.items():
        ps = [torch.nn.Parameter(torch.from_numpy(g["w0_%d" % i].copy())) for i in range(2)]
        cls = (optimizers.SharedRMSpropEpsInsideSqrt if name == "momentum"
               else optimizers.RMSpropEpsInsideSqrt)
        opt = cls(ps, lr=7e-4, eps=1e-1, alpha=0.99, **kw)
        if name == "momentum":      # state exists before the first step
            assert set(opt.state[ps[0]]) == {"step", "square_avg", "momentum_buffer"}
        for t in range(4):
            for i, p in enumerate(ps):
                p.grad = torch.from_numpy(g["g%d_%d" % (t, i)].copy())
            opt.step()
            for i, p in enumerate(ps):
                np.testing.assert_allclose(p.detach().numpy(), g["%s_w%d_%d" % (name, t, i)],
                                           rtol=1e-6, atol=1e-7, err_msg="%s step %d" % (name, t))
        for i, p in enumerate(ps):
            keys = {k[len(name) + 7:-2] for k in g.files if k.startswith(name + "_state_")}
            assert set(opt.state[p]) == keys
            for key in keys:
                np.testing.assert_allclose(np.asarray(opt.state[p][key]),
                                           g["%s_state_%s_%d" % (name, key, i)], rtol=1e-6,
                                           atol=1e-7)
    # eps inside the root: a zero-gradient-history parameter moves by lr * g / sqrt(v + eps)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = optimizers.RMSpropEpsInsideSqrt([p], lr=1.0, alpha=0.0, eps=3.0)
    p.grad = torch.ones(1)
    opt.step()
    assert float(p.detach()) == -0.5
