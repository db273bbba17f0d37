"""FusedRMSprop.step_from_sources (csrc/optim.hip k_rmsprop_fused): the optimizer step that takes
gradients as the backward pass left them -- split-K slabs, the hidden layer's batch matrices --
against torch.optim.RMSprop fed the materialised gradients (reference: the ``optimizer.step()``
of pfrl/agents/dqn.py:360-365 with examples/atari/train_dqn_batch_ale.py:199-206's RMSprop)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("centered", [True, False])
def test_step_from_sources_equals_torch_rmsprop(centered):
    from pfrl_amd.optimizers import FusedRMSprop, GradSource

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M, F, K = 32, 128, 192
    shapes = dict(plain=(300,), slabs=(64, 33), w=(F, K), b=(F,))
    params = {k: torch.nn.Parameter(torch.randn(s, device=dev)) for k, s in shapes.items()}
    ref_params = {k: torch.nn.Parameter(p.detach().clone()) for k, p in params.items()}
    opt = FusedRMSprop(list(params.values()), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=centered)
    ref = torch.optim.RMSprop(list(ref_params.values()), lr=2.5e-4, alpha=0.95, eps=1e-2,
                              centered=centered)
    for it in range(3):
        g_plain = torch.randn(shapes["plain"], device=dev)
        S, n = 5, 64 * 33
        stride = n + 40
        part = torch.randn(S * stride, device=dev)
        dy = torch.randn(M, F, device=dev)
        out = torch.randn(M, F, device=dev)          # the layer's ReLU output: mask = out > 0
        x = torch.randn(M, K, device=dev)
        fold_part = torch.randn(4 * 7, device=dev)
        fold_out = torch.empty(3, device=dev)
        params["plain"].grad = g_plain
        sources = {params["slabs"]: GradSource.slabs(part, stride, S),
                   params["w"]: GradSource.lowrank(dy, out, x),
                   params["b"]: GradSource.lowrank_bias(dy, out)}
        opt.step_from_sources(sources, folds=[(fold_part, fold_out, 7, 4)])
        # the same gradients, materialised (f64 products rounded to f32)
        dym = (dy * (out > 0)).double()
        ref_params["plain"].grad = g_plain
        ref_params["slabs"].grad = part.view(S, stride)[:, :n].sum(0).view(64, 33)
        ref_params["w"].grad = (dym.t() @ x.double()).float()
        ref_params["b"].grad = dym.sum(0).float()
        ref.step()
        for k in params:
            a, b = params[k].detach(), ref_params[k].detach()
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (it, k)
            sa, sb = opt.state[params[k]]["square_avg"], ref.state[ref_params[k]]["square_avg"]
            assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max()) + 1e-9, (it, k)
        want = fold_part.view(4, 7)[:, :3].sum(0)
        assert float((fold_out - want).abs().max()) <= 1e-6


def test_lowrank_tile_layout_is_not_transposed():
    """Asymmetric operands (guide: a symmetric B passes a row/col swap): W moves by exactly
    lr * g / (sqrt(g^2 (1 - alpha)) + eps) with g = dy^T x known in closed form."""
    from pfrl_amd.optimizers import FusedRMSprop, GradSource

    dev = torch.device("cuda:0")
    M, F, K = 4, 64, 128
    w = torch.nn.Parameter(torch.zeros(F, K, device=dev))
    dy = torch.zeros(M, F, device=dev)
    x = torch.zeros(M, K, device=dev)
    dy[0] = torch.arange(F, device=dev, dtype=torch.float32) + 1        # g[co][kk] = (co+1)(kk+1)
    x[0] = torch.arange(K, device=dev, dtype=torch.float32) + 1
    dy[2, 5] = 3.0
    x[2, 77] = -2.0                                                      # + one off-grid term
    opt = FusedRMSprop([w], lr=1.0, alpha=0.0, eps=1.0, centered=False)
    opt.step_from_sources({w: GradSource.lowrank(dy, None, x)})
    g = torch.outer(dy[0], x[0])
    g[5, 77] += -6.0
    want = -g / (g.abs() + 1.0)
    assert torch.allclose(w.detach(), want, rtol=1e-6, atol=1e-7)


def _run_updates(fused, n_updates=6, fwd_fold=True, extra_hidden=False):
    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.initializers import init_chainer_default
    from pfrl_amd.optimizers import FusedRMSprop
    from pfrl_amd.q_functions import DiscreteActionValueHead

    os.environ["PFRL_FUSED_OPT"] = "1" if fused else "0"
    os.environ["PFRL_FWD_FOLD"] = "1" if fwd_fold else "0"
    try:
        dev = torch.device("cuda:0")
        pfrl.utils.set_random_seed(0)
        torch.manual_seed(0)
        mid = [init_chainer_default(torch.nn.Linear(512, 512)), torch.nn.ReLU()] if extra_hidden else []
        q = torch.nn.Sequential(pfrl.nn.LargeAtariCNN(), *mid,
                                init_chainer_default(torch.nn.Linear(512, 6)),
                                DiscreteActionValueHead()).to(memory_format=torch.channels_last)
        opt = FusedRMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
        N = 32
        store = DeviceFrameStore(4096, (84, 84), torch.uint8, dev, stack=4)
        env = SyntheticAtariVectorEnv(N, store=store, seed=1, n_actions=6)
        rbuf = replay_buffers.ReplayBuffer(2000)
        ex = explorers.ConstantEpsilonGreedy(0.2, lambda: np.random.randint(6))
        ag = agents.DQN(q, opt, rbuf, gpu=0, gamma=0.99, explorer=ex, replay_start_size=200,
                        target_update_interval=10 ** 4, update_interval=4, minibatch_size=32,
                        batch_accumulator="sum", phi=lambda x: np.asarray(x, dtype=np.float32) / 255)
        obss = env.reset()
        while ag.optim_t < n_updates:
            a = ag.batch_act(obss)
            obss, r, d, _ = env.step(a)
            ag.batch_observe(obss, r, d, np.zeros(N, dtype=bool))
            obss = env.reset(~d)
        torch.cuda.synchronize()
        used = ag._graphed is not None and any(
            k[0] == "range" for k in ag._graphed.graphs)
        return ([p.detach().cpu().numpy().copy() for p in q.parameters()],
                np.asarray(ag.loss_record.values()), used)
    finally:
        os.environ.pop("PFRL_FUSED_OPT", None)
        os.environ.pop("PFRL_FWD_FOLD", None)


def test_dqn_updates_with_the_gradient_finishing_optimizer_match_the_plain_step():
    """Nature-CNN DQN updates as bench.py runs them (range graphs): with the optimizer finishing
    the gradients (default) and with fold + gradient tensors + plain step (PFRL_FUSED_OPT=0):
    same losses and parameters to f32 summation-order tolerance."""
    pa, la, ua = _run_updates(True)
    pb, lb, ub = _run_updates(False)
    assert ua and ub and len(la) == len(lb) >= 6
    np.testing.assert_allclose(la, lb, rtol=2e-5, atol=1e-6)
    for a, b in zip(pa, pb):
        assert np.abs(a - b).max() <= 2e-5 * max(1e-3, np.abs(b).max())


def test_hidden_layer_fold_inside_the_head_launch_is_bit_identical():
    """The hidden layer's split-K slabs folded by the head + TD-loss launch (default) vs by their
    own pfrl_splitk_reduce launch (PFRL_FWD_FOLD=0): same summation order, so the same bits --
    losses and every parameter after 6 updates."""
    pa, la, ua = _run_updates(True, fwd_fold=True)
    pb, lb, ub = _run_updates(True, fwd_fold=False)
    assert ua and ub
    np.testing.assert_array_equal(la, lb)
    for a, b in zip(pa, pb):
        np.testing.assert_array_equal(a, b)


def test_module_between_trunk_and_head_reads_the_folded_hidden_layer():
    """Sequential(cnn, Linear(512, 512), ReLU, Linear(512, A), head): the trunk's split-K slabs
    may only be handed to the narrow head's launch; the layer in between must see the folded
    tensor (it read an unfilled buffer).  Same bits with and without the sink."""
    pa, la, ua = _run_updates(True, fwd_fold=True, extra_hidden=True)
    pb, lb, ub = _run_updates(True, fwd_fold=False, extra_hidden=True)
    assert ua and ub and np.all(np.isfinite(la))
    np.testing.assert_array_equal(la, lb)
    for a, b in zip(pa, pb):
        np.testing.assert_array_equal(a, b)


def test_optimizer_steps_riding_in_the_backward_launches_are_bit_identical():
    """RMSprop steps as extra workgroups of the backward launches against the optimizer's own
    launch (same arithmetic on the same gradients / the same slab sums in the same order, so the
    same bits -- losses and every parameter after 6 updates):
      off    PFRL_RIDE_ALONG=0: every step in pfrl_rmsprop_fused_step
      last   round 5: the hidden layer's step in the first convolution's weight-gradient launch
      all    round 6 (PFRL_RIDE_MORE=1; measured slower, not the default): + the head's in conv3's
             backward launch, conv3's in conv2's, conv2's in conv1's (pfrl_ride_set); the
             optimizer launch keeps conv1 + the loss fold
    and the rides are really taken."""
    from pfrl_amd import _native
    from pfrl_amd.nn import mfma_trunk

    calls = {"ride": 0, "set": 0}
    lib = _native.lib()
    real_ride, real_set = lib.pfrl_conv2d_nhwc_bwd_weight_ride, lib.pfrl_ride_set

    class _Spy:
        def __getattr__(self, name):
            if name == "pfrl_conv2d_nhwc_bwd_weight_ride":
                def f(*a):
                    calls["ride"] += 1
                    return real_ride(*a)
                return f
            if name == "pfrl_ride_set":
                def f(*a):
                    calls["set"] += 1 if a[0] > 0 else 0
                    return real_set(*a)
                return f
            return getattr(lib, name)

    old_lib, old_ride, old_more = _native.lib, mfma_trunk._RIDE, mfma_trunk._RIDE_MORE
    _native.lib = lambda: _Spy()
    mfma_trunk._native.lib = _native.lib
    from pfrl_amd import optimizers

    optimizers._native.lib = _native.lib
    out = {}
    try:
        for name, ride, more in (("all", True, True), ("last", True, False), ("off", False, False)):
            mfma_trunk._RIDE, mfma_trunk._RIDE_MORE = ride, more
            calls["ride"] = calls["set"] = 0
            p, l, used = _run_updates(True)
            out[name] = (p, l, used, dict(calls))
    finally:
        _native.lib = old_lib
        mfma_trunk._native.lib = old_lib
        optimizers._native.lib = old_lib
        mfma_trunk._RIDE, mfma_trunk._RIDE_MORE = old_ride, old_more
    assert all(o[2] for o in out.values())
    assert out["all"][3]["ride"] >= 1 and out["all"][3]["set"] >= 3 * out["all"][3]["ride"]
    assert out["last"][3]["ride"] >= 1 and out["last"][3]["set"] == 0
    assert out["off"][3] == {"ride": 0, "set": 0}
    for name in ("last", "off"):
        np.testing.assert_array_equal(out["all"][1], out[name][1])
        for a, b in zip(out["all"][0], out[name][0]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_data_parallel_update_keeps_the_fused_forms(backend, tmp_path, monkeypatch):
    """VERDICT r3 item 3(a, b): under a process group (single rank here) the captured update keeps
    the fused head + TD-loss launch and the split-K slabs -- folded straight into the flat bucket by
    ONE launch (GradientAllReducer.pack_sources) -- and the hidden layer's gradient is exchanged as
    its batch matrices (all-gather of dy and x, local product; forced at world size 1).  nccl: the
    collectives are captured inside the range graph after the capture probe; gloo: graph -> eager
    collective -> graph.  Same losses and parameters as the single-process run to f32
    summation-order tolerance, and the low-rank path is really taken."""
    import torch.distributed as dist

    from pfrl_amd import distributed

    ref_p, ref_l, used = _run_updates(True)
    assert used and not dist.is_initialized()
    kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="file://%s" % (tmp_path / "pg"), rank=0,
                            world_size=1, **kw)
    monkeypatch.setenv("PFRL_FORCE_SPLIT_GRAPH", "1")
    monkeypatch.setenv("PFRL_DP_LOWRANK", "force")
    # nccl: the directly driven RCCL communicator (pfrl_amd/rccl.py); gloo: the process group itself
    monkeypatch.setenv("PFRL_RCCL_DIRECT", "1" if backend == "nccl" else "0")
    taken = []
    orig = distributed.GradientAllReducer.lowrank_ready

    def spy(self, *a, **kw):
        r = orig(self, *a, **kw)
        taken.append(r)
        return r

    monkeypatch.setattr(distributed.GradientAllReducer, "lowrank_ready", spy)
    packed = []
    orig_pack = distributed.GradientAllReducer.pack_sources

    def spy_pack(self, sources):
        n = len(sources)
        orig_pack(self, sources)
        packed.append((n, len(sources)))

    monkeypatch.setattr(distributed.GradientAllReducer, "pack_sources", spy_pack)
    try:
        if backend == "nccl":
            assert distributed.captured_collectives_work(torch.device("cuda:0"))
        pa, la, ua = _run_updates(True)
    finally:
        dist.destroy_process_group()
    # (the split plan's captures leave collectives out: there only the eager warm-up takes it)
    assert taken and (all(taken) if backend == "nccl" else any(taken)), "low-rank exchange not taken"
    # eight slab sources (three convolutions + the head, weight and bias) folded into the bucket
    assert packed and all(before >= 8 and after == before - 8 for before, after in packed), packed
    assert ua == (backend == "nccl")      # range graphs only with the collective captured
    assert len(la) == len(ref_l) >= 6
    np.testing.assert_allclose(la, ref_l, rtol=2e-5, atol=1e-6)
    for a, b in zip(pa, ref_p):
        assert np.abs(a - b).max() <= 2e-5 * max(1e-3, np.abs(b).max())
