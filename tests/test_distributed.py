"""Multi-process (world_size = 2, gloo, CPU) tests of the env-sharded data
parallel path: one flat gradient all-reduce per update, env sharding,
parameter broadcast.  The same code runs over RCCL/xGMI with backend 'nccl'."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from pfrl_amd import distributed

    r, w, _ = distributed.init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    assert distributed.shard_envs(256) == (rank * 128, (rank + 1) * 128)
    torch.manual_seed(100 + rank)   # different init per rank on purpose
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    red = distributed.GradientAllReducer(model)
    red.broadcast_parameters(model, src=0)
    # every rank sees its own shard of one global batch
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    loss = torch.nn.functional.mse_loss(model(xs), ys, reduction="sum")
    model.zero_grad()
    loss.backward()
    red.all_reduce()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # PPO advantage statistics over the union of the shards
    adv = torch.randn(1000 + 37 * rank, generator=torch.Generator().manual_seed(50 + rank)) * (
        1 + rank) + 0.3 * rank
    std, mean = torch.std_mean(adv, unbiased=False)
    gms = distributed.global_mean_std(torch.stack([mean, std]), adv.numel())
    np.save(os.path.join(out_dir, "gms%d.npy" % rank), gms.numpy())
    # observation statistics over the union of the shards, batch after batch
    from pfrl_amd.nn import EmpiricalNormalization

    norm = EmpiricalNormalization(5, clip_threshold=5)
    gen = torch.Generator().manual_seed(90)
    batches = [torch.randn(6 + 4 * k, 5, generator=gen) * (1 + k) + k for k in range(3)]
    for b in batches:
        half = b.shape[0] // 2 + 1      # unequal shards
        norm(b[:half] if rank == 0 else b[half:])
    np.save(os.path.join(out_dir, "norm%d.npy" % rank),
            torch.cat([norm.mean, norm.std, norm.count.float().reshape(1)]).numpy())
    if rank == 0:
        single = EmpiricalNormalization(5, clip_threshold=5)
        single.sync_across_ranks = False
        for b in batches:
            single(b)
        np.save(os.path.join(out_dir, "norm_ref.npy"),
                torch.cat([single.mean, single.std, single.count.float().reshape(1)]).numpy())
    np.save(os.path.join(out_dir, "adv%d.npy" % rank), adv.numpy())
    np.save(os.path.join(out_dir, "grad%d.npy" % rank), flat.numpy())
    np.save(os.path.join(out_dir, "param%d.npy" % rank), params.numpy())
    if rank == 0:
        # single-process reference: mean over ranks of the shard gradients
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        ref.load_state_dict(model.state_dict())
        total = torch.nn.functional.mse_loss(ref(x), y, reduction="sum") / world
        ref.zero_grad()
        total.backward()
        np.save(os.path.join(out_dir, "ref.npy"),
                torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).numpy())
    dist.barrier()
    dist.destroy_process_group()


def _early_worker(rank, world, port, out_dir):
    """Split plan of GradientAllReducer: the large tensor's all-reduce starts as soon as its
    gradient exists (post-accumulate hook, or the producer's announce_grad), the rest goes
    in the flat bucket; must equal the single-bucket plan bit for bit."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import copy

    import torch.distributed as dist

    from pfrl_amd import distributed

    distributed.init_process_group_from_env(backend="gloo")
    torch.manual_seed(5)
    model = torch.nn.Sequential(torch.nn.Linear(16, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4096),
                                torch.nn.ReLU(), torch.nn.Linear(4096, 3))
    twin = copy.deepcopy(model)
    twin2 = copy.deepcopy(model)
    single = distributed.GradientAllReducer(twin, early_bytes=0)
    split = distributed.GradientAllReducer(model, early_bytes=100_000)      # 8 x 4096 x 4 B = 128 KB
    manual = distributed.GradientAllReducer(twin2, early_bytes=40_000)      # both wide layers
    assert len(split._early) == 1 and len(single._early) == 0 and len(manual._early) == 2
    for h in manual._hooks:      # this one is driven by the producer's announcement only
        h.remove()
    x = torch.randn(6, 16, generator=torch.Generator().manual_seed(10 + rank))
    started = []
    orig = split._start
    split._start = lambda t: (started.append(tuple(t.shape)), orig(t))[1]
    for m in (model, twin, twin2):
        m.zero_grad(set_to_none=True)
        m(x).pow(2).sum().backward()
    assert started == [(4096, 8)]            # launched from inside backward, by the hook
    for p in twin2.parameters():
        distributed.announce_grad(p, p.grad)  # what a fused backward node does per tensor
    assert len(manual._pending) == 2
    for r in (single, split, manual):
        r.all_reduce()
        assert not r._pending
    a, b, c = (torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in (twin, model, twin2))
    assert torch.equal(a, b) and torch.equal(a, c)
    np.save(os.path.join(out_dir, "early%d.npy" % rank), b.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_early_all_reduce_of_the_large_gradient_equals_single_bucket(tmp_path):
    port = _free_port()
    mp.spawn(_early_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = (np.load(os.path.join(str(tmp_path), "early%d.npy" % r)) for r in (0, 1))
    assert np.array_equal(g0, g1) and np.abs(g0).sum() > 0


def _lowrank_worker(rank, world, port, out_dir):
    """Low-rank exchange of a large Linear layer's gradient (all-gather of dy and x, local
    dW = sum_g dy_g^T x_g / G) against the flat all-reduce plan and against ONE process on the
    concatenated batch: 20 RMSprop updates, parameters compared after each plan."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import copy

    import torch.distributed as dist

    from pfrl_amd import distributed

    distributed.init_process_group_from_env(backend="gloo")
    torch.manual_seed(5)
    base = torch.nn.Sequential(torch.nn.Linear(16, 256), torch.nn.ReLU(), torch.nn.Linear(256, 512),
                               torch.nn.ReLU(), torch.nn.Linear(512, 3))
    M = 4
    assert distributed.lowrank_pays(M, 512, 256, world) and not distributed.lowrank_pays(4096, 512, 256, world)
    nets = {k: copy.deepcopy(base) for k in ("lowrank", "flat", "one")}
    red = {"lowrank": distributed.GradientAllReducer(nets["lowrank"], early_bytes=100_000),
           "flat": distributed.GradientAllReducer(nets["flat"], early_bytes=0)}
    assert len(red["lowrank"]._early) == 1 and len(red["lowrank"]._lowrank_modules) == 1
    opts = {k: torch.optim.RMSprop(n.parameters(), lr=1e-3, alpha=0.95, eps=1e-2, centered=True)
            for k, n in nets.items()}
    taken = []
    orig = red["lowrank"].lowrank_ready
    red["lowrank"].lowrank_ready = lambda *a: (taken.append(orig(*a)), taken[-1])[1]
    started = []
    orig_start = red["lowrank"]._start
    red["lowrank"]._start = lambda t: (started.append(tuple(t.shape)), orig_start(t))[1]
    for step in range(20):
        gen = torch.Generator().manual_seed(1000 + step)
        xs = torch.randn(world * M, 16, generator=gen)
        ys = torch.randn(world * M, 3, generator=gen)
        mine = slice(rank * M, (rank + 1) * M)
        for k in ("lowrank", "flat"):
            opts[k].zero_grad(set_to_none=True)
            # (sum over the shard, averaged over ranks by the reducer = mean over ranks of shard sums)
            torch.nn.functional.mse_loss(nets[k](xs[mine]), ys[mine], reduction="sum").backward()
            red[k].all_reduce()
            assert not red[k]._pending and not red[k]._lowrank
            opts[k].step()
        opts["one"].zero_grad(set_to_none=True)
        (torch.nn.functional.mse_loss(nets["one"](xs), ys, reduction="sum") / world).backward()
        opts["one"].step()
    assert taken == [True] * 20 and started == []     # never the all-reduce of the 512 x 256 gradient
    flat = {k: torch.cat([p.detach().reshape(-1) for p in n.parameters()]) for k, n in nets.items()}
    scale = float(flat["one"].abs().max())
    assert float((flat["lowrank"] - flat["flat"]).abs().max()) <= 1e-6 * scale
    assert float((flat["lowrank"] - flat["one"]).abs().max()) <= 1e-6 * scale
    np.save(os.path.join(out_dir, "lowrank%d.npy" % rank), flat["lowrank"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_lowrank_all_gather_of_a_large_linear_gradient_equals_the_flat_all_reduce(tmp_path):
    """parameters after 20 updates are those of the flat all-reduce plan and
    of a single process on the concatenated batch (<= 1e-6), identical on both ranks, and the
    large gradient itself never crosses the link."""
    port = _free_port()
    mp.spawn(_lowrank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = (np.load(os.path.join(str(tmp_path), "lowrank%d.npy" % r)) for r in (0, 1))
    assert np.array_equal(g0, g1) and np.abs(g0).sum() > 0


def test_gradient_all_reduce_two_ranks_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(tmp_path / "grad0.npy")
    g1 = np.load(tmp_path / "grad1.npy")
    ref = np.load(tmp_path / "ref.npy")
    np.testing.assert_array_equal(g0, g1)            # identical on every rank
    np.testing.assert_allclose(g0, ref, rtol=1e-5, atol=1e-6)   # == 1-process run on the full batch
    np.testing.assert_array_equal(np.load(tmp_path / "param0.npy"),
                                  np.load(tmp_path / "param1.npy"))   # broadcast worked
    np.testing.assert_array_equal(np.load(tmp_path / "norm0.npy"), np.load(tmp_path / "norm1.npy"))
    np.testing.assert_allclose(np.load(tmp_path / "norm0.npy"), np.load(tmp_path / "norm_ref.npy"),
                               rtol=1e-5, atol=1e-6)    # == one process fed the concatenated batches
    allv = np.concatenate([np.load(tmp_path / "adv0.npy"), np.load(tmp_path / "adv1.npy")])
    for r in range(2):
        gms = np.load(tmp_path / ("gms%d.npy" % r))
        np.testing.assert_allclose(gms, [allv.mean(), allv.std()], rtol=1e-5)


def test_shard_envs_single_process():
    from pfrl_amd import distributed

    assert distributed.world_size() == 1
    assert distributed.shard_envs(256) == (0, 256)
    assert distributed.shard_envs(512, rank=3, world=8) == (192, 256)
    with pytest.raises(AssertionError):
        distributed.shard_envs(10, rank=0, world=4)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("prioritized", [False, True])
def test_data_parallel_graph_plans_match_reference(prioritized, backend, tmp_path):
    """The data-parallel forms of the captured update under a real (single-rank)
    process group reproduce the reference trace:
      gloo: graph(fwd + bwd + pack) -> eager all-reduce -> graph(unpack + step)
      nccl: the RCCL all-reduce is captured INSIDE the graph (one replay per update)
    and with PER the forward pass is its own graph in front (replay-stream hand-over)."""
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_agent_parity as T

    assert not dist.is_initialized()
    kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="file://%s" % (tmp_path / "pg"), rank=0,
                            world_size=1, **kw)
    os.environ["PFRL_FORCE_SPLIT_GRAPH"] = "1"
    os.environ["PFRL_GRAPH_COLLECTIVE"] = "1"    # captured collective where the backend allows
    # gloo: the process group itself carries the gradients (host round trip, eager, between two
    # graphs); nccl: the data plane is the directly driven RCCL communicator, inside the graph
    os.environ["PFRL_RCCL_DIRECT"] = "0" if backend == "gloo" else "1"
    try:
        if prioritized:
            g = np.load(os.path.join(T.GOLDEN, "agent_trace_ddqn_per_n3.npz"))
            got = T._run("ddqn", True, 3, True, gpu=0)
        else:
            g = np.load(os.path.join(T.GOLDEN, "agent_trace_dqn_uniform_n1.npz"))
            got = T._run("dqn", False, 1, False, gpu=0)
        ag = got["agent"]
        assert ag._graphed is not None and ag._graphed.split_for_allreduce
        entries = list(ag._graphed.graphs.values())
        plans = [e["plan"] for e in entries if "plan" in e]
        assert entries
        if backend == "nccl":
            assert ag._graphed.graph_collective not in ("0", False)
            assert all("all_reduce" not in p for p in plans)
            assert all(len(p) == (3 if prioritized else 1) for p in plans)
            if not prioritized:
                # with the collective inside the graph a whole env range replays as one graph
                assert any("graph" in e for e in entries)
        else:
            assert all("all_reduce" in p for p in plans)
        assert ag.grad_reducer._flat is not None
        T._compare(got, g)
    finally:
        os.environ.pop("PFRL_FORCE_SPLIT_GRAPH", None)
        os.environ.pop("PFRL_GRAPH_COLLECTIVE", None)
        os.environ.pop("PFRL_RCCL_DIRECT", None)
        dist.destroy_process_group()


@pytest.mark.gpu
def test_ppo_captured_update_with_the_gradient_all_reduce_inside_equals_one_process(tmp_path, monkeypatch):
    """PPO's captured minibatch update (agents/ppo.py::_minibatch_step) under a real single-rank
    process group with the directly driven RCCL communicator: the gradient all-reduce is a node
    of the captured graph (no eager fallback), the 3-scalar advantage statistics go through the
    control plane, and two rollouts leave the parameters of the run WITHOUT a process group, bit
    for bit (an average over one rank is the identity)."""
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from test_bench_path_parity import _bench_args

    dev = torch.device("cuda:0")
    N, T = 32, 16

    def run():
        args = _bench_args(algo="ppo", num_envs=N)
        agent, env, _ = bench.build_agent(args, dev, 0)
        agent.update_interval = N * T
        agent.minibatch_size = N * T // 4
        obss = env.reset()
        for _ in range(2 * T):
            obss = bench.one_step(agent, env, obss, N)
        torch.cuda.synchronize()
        assert agent.n_updates == 2 * 4 * 4
        assert agent._update_graph is not None and len(agent._update_graph.graphs) >= 1, \
            "the captured update was not taken"
        return agent, [p.detach().cpu().clone() for p in agent.model.parameters()]

    assert not dist.is_initialized()
    _, want = run()
    monkeypatch.setenv("PFRL_DIST_ALWAYS", "1")
    monkeypatch.setenv("PFRL_RCCL_DIRECT", "1")
    dist.init_process_group("gloo", init_method="file://%s" % (tmp_path / "pg"), rank=0, world_size=1)
    try:
        agent, got = run()
        assert agent.grad_reducer.active() and agent.grad_reducer._comm is not None
        assert agent.grad_reducer.current_bucket() is not None, "no gradient went through the all-reduce"
        for a, b in zip(want, got):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def _agent_worker(rank, world, port, out_dir, kind):
    """Each rank trains on its own env shard (different env seeds, different replay
    contents); after the broadcast of rank 0's initial weights and with one averaged
    gradient per update, the replicas must stay bit-identical."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import tempfile

    import torch.distributed as dist

    import pfrl_amd as pfrl
    from pfrl_amd import agents, distributed, explorers, replay_buffers
    from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv, HostSyntheticVectorObsEnv

    distributed.init_process_group_from_env(backend="gloo")
    torch.set_num_threads(1)
    pfrl.utils.set_random_seed(100 + rank)     # different init and different exploration per rank
    lo, hi = distributed.shard_envs(8)
    n_local = hi - lo
    if kind == "dqn":
        env = HostSyntheticAtariVectorEnv(n_local, seed=10 + rank, frame_shape=(8, 8), p_done=0.05)
        q = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4 * 64, 16), torch.nn.ReLU(),
                                torch.nn.Linear(16, 6), pfrl.q_functions.DiscreteActionValueHead())
        opt = torch.optim.SGD(q.parameters(), lr=1e-2)
        ag = agents.DQN(q, opt, replay_buffers.ReplayBuffer(500), 0.99,
                        explorers.ConstantEpsilonGreedy(0.3, lambda: np.random.randint(6)),
                        gpu=-1, replay_start_size=20, minibatch_size=8, update_interval=4,
                        target_update_interval=40,
                        phi=lambda x: np.asarray(x, dtype=np.float32) / 255)
        ag.grad_reducer.broadcast_parameters(ag.model)
        ag.sync_target_network()
        nets = [q]
    elif kind in ("ppo", "a2c"):
        env = HostSyntheticAtariVectorEnv(n_local, seed=30 + rank, frame_shape=(8, 8), p_done=0.05)
        model = torch.nn.Sequential(
            torch.nn.Flatten(), torch.nn.Linear(4 * 64, 16), torch.nn.ReLU(),
            pfrl.nn.Branched(torch.nn.Sequential(torch.nn.Linear(16, 6),
                                                 pfrl.policies.SoftmaxCategoricalHead()),
                             torch.nn.Linear(16, 1)))
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        phi = lambda x: np.asarray(x, dtype=np.float32) / 255     # noqa: E731
        if kind == "ppo":
            ag = agents.PPO(model, opt, gpu=-1, phi=phi, update_interval=32, minibatch_size=8,
                            epochs=2, standardize_advantages=True, max_grad_norm=0.5)
            seen = []
            real = ag._host._advantage_statistics
            ag._host._advantage_statistics = lambda tr: (seen.append(real(tr)) or seen[-1])
        else:
            ag = agents.A2C(model, opt, gamma=0.99, num_processes=n_local, gpu=-1, update_steps=5,
                            phi=phi, max_grad_norm=0.5)
        distributed.broadcast_agent(ag)
        nets = [model]
    elif kind == "sac":
        obs_dim, act_dim = 10, 2
        env = HostSyntheticVectorObsEnv(n_local, obs_dim=obs_dim, act_dim=act_dim, seed=20 + rank,
                                        p_done=0.05)
        policy = torch.nn.Sequential(
            torch.nn.Linear(obs_dim, 16), torch.nn.ReLU(), torch.nn.Linear(16, act_dim),
            pfrl.policies.GaussianHeadWithFixedCovariance(0.3))
        mkq = lambda: torch.nn.Sequential(pfrl.nn.ConcatObsAndAction(),
                                          torch.nn.Linear(obs_dim + act_dim, 16),
                                          torch.nn.ReLU(), torch.nn.Linear(16, 1))
        q1, q2 = mkq(), mkq()
        opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
        ag = agents.SoftActorCritic(
            policy, q1, q2, opts[0], opts[1], opts[2], replay_buffers.ReplayBuffer(500),
            gamma=0.99, gpu=-1, replay_start_size=20, minibatch_size=8, update_interval=1,
            entropy_target=-float(act_dim), temperature_optimizer_lr=1e-2,
            initial_temperature=1.0 + rank,       # replicas start apart on purpose
            burnin_action_func=lambda: np.random.uniform(-1, 1, act_dim).astype(np.float32))
        distributed.broadcast_agent(ag)           # every saved module, targets and temperature
        nets = [policy, q1, q2, ag.target_q_func1, ag.target_q_func2, ag.temperature_holder]
    else:
        obs_dim, act_dim = 10, 2
        env = HostSyntheticVectorObsEnv(n_local, obs_dim=obs_dim, act_dim=act_dim, seed=20 + rank,
                                        p_done=0.05)
        policy = torch.nn.Sequential(
            torch.nn.Linear(obs_dim, 16), torch.nn.ReLU(), torch.nn.Linear(16, act_dim),
            pfrl.nn.BoundByTanh(low=-np.ones(act_dim, dtype=np.float32),
                                high=np.ones(act_dim, dtype=np.float32)),
            pfrl.policies.DeterministicHead())
        mkq = lambda: torch.nn.Sequential(pfrl.nn.ConcatObsAndAction(),
                                          torch.nn.Linear(obs_dim + act_dim, 16),
                                          torch.nn.ReLU(), torch.nn.Linear(16, 1))
        q1, q2 = mkq(), mkq()
        opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (policy, q1, q2)]
        ag = agents.TD3(policy, q1, q2, opts[0], opts[1], opts[2], replay_buffers.ReplayBuffer(500),
                        gamma=0.99, explorer=explorers.AdditiveGaussian(0.1, -1.0, 1.0), gpu=-1,
                        replay_start_size=20, minibatch_size=8, update_interval=1,
                        burnin_action_func=lambda: np.random.uniform(-1, 1, act_dim).astype(
                            np.float32),
                        target_policy_smoothing_func=lambda a: torch.clamp(a + 0.05, -1, 1))
        nets = [policy, q1, q2]
        for m in nets:
            ag._reducers[m].broadcast_parameters(m)
        for src, dst in ((policy, ag.target_policy), (q1, ag.target_q_func1),
                         (q2, ag.target_q_func2)):
            dst.load_state_dict(src.state_dict())
    pfrl.experiments.train_agent_batch(ag, env, 200 * n_local // 4, tempfile.mkdtemp())
    flat = np.concatenate([p.detach().numpy().ravel() for m in nets for p in m.parameters()])
    n_updates = {"dqn": lambda: ag.optim_t, "td3": lambda: ag.q_func_n_updates,
                 "sac": lambda: ag.n_policy_updates, "ppo": lambda: ag.n_updates,
                 "a2c": lambda: ag.t // 5}[kind]()
    if kind == "ppo":      # advantage statistics are those of the union of the shards
        np.save(os.path.join(out_dir, "ppo_advstats%d.npy" % rank),
                np.asarray([[float(m), float(s)] for m, s in seen]))
    np.save(os.path.join(out_dir, "%s_params%d.npy" % (kind, rank)), flat)
    np.save(os.path.join(out_dir, "%s_updates%d.npy" % (kind, rank)), np.asarray(n_updates))
    np.save(os.path.join(out_dir, "%s_rlen%d.npy" % (kind, rank)),
            np.asarray(len(ag.replay_buffer) if hasattr(ag, "replay_buffer") else 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["dqn", "td3", "sac", "ppo", "a2c"])
def test_env_sharded_agents_stay_in_sync_two_ranks_gloo(tmp_path, kind):
    """world_size 2, gloo, host replay: env shards and replay contents differ per rank,
    the replicas do not (one averaged gradient per optimizer step)."""
    world = 2
    mp.spawn(_agent_worker, args=(world, _free_port(), str(tmp_path), kind), nprocs=world,
             join=True)
    p0 = np.load(tmp_path / ("%s_params0.npy" % kind))
    p1 = np.load(tmp_path / ("%s_params1.npy" % kind))
    u0 = int(np.load(tmp_path / ("%s_updates0.npy" % kind)))
    u1 = int(np.load(tmp_path / ("%s_updates1.npy" % kind)))
    assert u0 == u1 >= 10
    np.testing.assert_array_equal(p0, p1)
    assert np.isfinite(p0).all()
    if kind == "ppo":
        s0, s1 = np.load(tmp_path / "ppo_advstats0.npy"), np.load(tmp_path / "ppo_advstats1.npy")
        assert len(s0) > 3
        np.testing.assert_array_equal(s0, s1)
