"""csrc/qnet.hip -- the f32 MFMA trunk kernels -- against the same ops in stock PyTorch fp32
(CPU), through the C ABI; plus the host-side plumbing that routes models onto them.

Tolerances: the kernels accumulate in f32 like PyTorch does, in a different order; every
check is relative to the largest magnitude of the reference tensor."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import pfrl_amd
from pfrl_amd.nn import mfma_trunk as mt

gpu = pytest.mark.gpu


def _close(a, b, rel):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max()) / scale
    assert err <= rel, "relative error %.3e > %.1e" % (err, rel)


def _nature_q(n_actions=6):
    torch.manual_seed(0)
    return nn.Sequential(pfrl_amd.nn.LargeAtariCNN(), nn.Linear(512, n_actions))


# ---------------------------------------------------------------------------- host logic
def test_trunk_plan_is_none_off_the_device():
    m = pfrl_amd.nn.LargeAtariCNN()
    x = torch.zeros(2, 4, 84, 84)
    assert mt.plan_for(m.layers, m.output, x) is None
    assert m(x).shape == (2, 512)       # stock route


def test_sequential_trunk_run_detection_and_state_dict():
    seq = nn.Sequential(nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2),
                        nn.ReLU(), nn.Conv2d(64, 64, 3), nn.ReLU(), nn.Flatten(),
                        nn.Linear(3136, 512), nn.ReLU(), nn.Linear(512, 7))
    keys = list(seq.state_dict().keys())
    x = torch.rand(3, 4, 84, 84)
    want = seq(x)
    pfrl_amd.nn.fuse_sequential_trunk(seq)
    pfrl_amd.nn.accelerate_heads(seq)
    assert type(seq).__name__ == "_TrunkSequential" and seq._trunk_run == (0, 9, [0, 2, 4], 7)
    assert list(seq.state_dict().keys()) == keys
    assert torch.equal(seq(x), want)    # CPU input: child by child, unchanged
    clone = copy.deepcopy(seq)
    assert clone._trunk_run == seq._trunk_run and torch.equal(clone(x), want)
    # a model without such a stretch is left alone
    mlp = nn.Sequential(nn.Linear(4, 8), nn.ReLU(), nn.Linear(8, 2))
    assert type(pfrl_amd.nn.fuse_sequential_trunk(mlp)) is nn.Sequential


def test_split_heuristics_cover_the_reduction():
    for M, F_, K in [(32, 512, 3136), (256, 512, 3136), (2048, 512, 3136), (5, 512, 3136)]:
        s = mt._fwd_splits(M, F_, K)
        cps = -(-(K // 32) // s)
        assert 1 <= s <= K // 32 and cps * s >= K // 32 and cps * (s - 1) < K // 32
    for M, Co, K in [(12800, 32, 256), (2592, 64, 512), (1568, 64, 576), (102400, 32, 256), (7, 64, 576),
                     (6553600, 32, 256), (1327104, 64, 512), (802816, 64, 576), (16384, 512, 3136)]:
        s = mt._wgrad_splits(M, Co, K)
        nch = -(-M // 32)
        cps = -(-nch // s)
        assert 1 <= s <= nch and cps * s >= nch and cps * (s - 1) < nch


# ---------------------------------------------------------------------------- kernels
GEOMS = [(4, 32, 8, 4, 84), (32, 64, 4, 2, 20), (64, 64, 3, 1, 9),      # Nature
         (4, 16, 8, 4, 84), (16, 32, 4, 2, 20)]                        # NIPS-2013 (16-wide tiles)


@gpu
@pytest.mark.parametrize("B", [32, 5])
@pytest.mark.parametrize("geom", GEOMS)
def test_conv_kernels_match_torch(geom, B):
    C, Co, R, ST, H = geom
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 100 + C)
    conv = nn.Conv2d(C, Co, R, stride=ST)
    x = torch.rand(B, C, H, H)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(conv(xr))
    gy = torch.randn_like(yr)
    yr.backward(gy)
    cg = copy.deepcopy(conv).to(dev).to(memory_format=torch.channels_last)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    sp = mt.ConvSpec(cg, H, H)
    y = mt.conv_fwd(xg, cg.weight, cg.bias, sp, B, relu=True, planar=False)
    _close(y.permute(0, 3, 1, 2), yr, 2e-6)
    yp = mt.conv_fwd(xg, cg.weight, cg.bias, sp, B, relu=True, planar=True)
    assert torch.equal(yp.view(B, Co, sp.OH, sp.OW), y.permute(0, 3, 1, 2))
    lib = mt._native.lib()
    dy = (gy * (yr > 0)).permute(0, 2, 3, 1).contiguous().to(dev)
    nW, M = cg.weight.numel(), B * sp.OH * sp.OW
    splits = mt._wgrad_splits(M, Co, R * R * C)
    stride = nW + Co
    part = torch.empty(max(splits, 1) * stride, device=dev)
    dw, db = torch.empty_like(cg.weight), torch.empty(Co, device=dev)
    mt.check(lib.pfrl_conv2d_nhwc_bwd_weight(mt._p(dy), None, mt._p(xg), mt._p(part), mt._p(part[nW:]),
                                              stride, stride, B, H, H, C, Co, R, R, ST, splits,
                                              mt._stream()), "wgrad")
    mt._reduce([(part, dw, None, stride, nW, splits, 4, 0), (part[nW:], db, None, stride, Co, splits, 4, 0)])
    assert dw.stride() == cg.weight.stride()
    _close(dw, conv.weight.grad, 5e-6)
    _close(db, conv.bias.grad, 5e-6)
    if C % 16 == 0:
        aprev = torch.rand(B, H, H, C, device=dev) - 0.3
        dx = torch.full((B, H, H, C), float("nan"), device=dev)
        mt.check(lib.pfrl_conv2d_nhwc_bwd_data(mt._p(dy), None, mt._p(cg.weight), mt._p(aprev), mt._p(dx),
                                               B, H, H, C, Co, R, R, ST, 0, 0, mt._stream()), "dgrad")
        _close(dx, xr.grad.permute(0, 2, 3, 1) * (aprev.cpu() > 0), 5e-6)


@gpu
def test_nhwc_route_of_the_hidden_layer_matches_torch_and_sees_raw_pointer_updates():
    """From 1 024 observations up the trunk's last convolution writes NHWC rows and the hidden layer
    reads a column-re-ordered copy of its weight (csrc-free: one torch copy per forward pass).
    Forward, every gradient, and -- the hazard a cached copy would have -- a forward pass AFTER the
    parameters were stepped by this package's own optimizer, which writes through raw pointers that
    a tensor's version counter never sees, against stock PyTorch on the same (updated) weights."""
    from pfrl_amd.optimizers import FusedRMSprop

    dev = torch.device("cuda:0")
    B = 1056                       # > 1 024 and ragged against every tile height
    ref = _nature_q()
    dut = copy.deepcopy(ref).to(dev).to(memory_format=torch.channels_last)
    mt.accelerate_heads(dut)
    torch.manual_seed(3)
    x = torch.rand(B, 4, 84, 84)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    assert B >= mt._NHWC_FC_MIN_BATCH and mt.plan_for(dut[0].layers, dut[0].output, xg) is not None
    q_ref, q = ref(x), dut(xg)
    _close(q, q_ref, 5e-6)
    g = torch.randn_like(q_ref)
    q_ref.backward(g)
    q.backward(g.to(dev))
    for (name, p), (_, pr) in zip(dut.named_parameters(), ref.named_parameters()):
        assert p.grad.stride() == p.stride(), name
        _close(p.grad, pr.grad, 2e-3)       # (B x 400 terms per conv1 weight, f32 on both sides)
    opt = FusedRMSprop(dut.parameters(), lr=1e-2, alpha=0.95, eps=1e-2, centered=True)
    opt_ref = torch.optim.RMSprop(ref.parameters(), lr=1e-2, alpha=0.95, eps=1e-2, centered=True)
    v0 = dut[0].output.weight._version
    opt.step()
    opt_ref.step()
    assert dut[0].output.weight._version == v0      # (the premise: the write is invisible to autograd)
    with torch.no_grad():
        moved = float((dut(xg) - q).abs().max())
        assert moved > 1e-3                          # the step really changed the function
        # and the trunk computes with the NEW weights.  (Tolerance: the two sides step on gradients
        # that agree to 2e-3 above, and a centred RMSprop step is lr * g / sqrt(var): the gradient
        # difference reaches the outputs undamped -- 2.1e-4 measured; a stale copy would be off by the
        # whole step, `moved` above.)
        _close(dut(xg), ref(x), 5e-4)


@gpu
@pytest.mark.parametrize("geom", [(4, 32, 8, 4, 84), (32, 64, 4, 2, 20), (64, 64, 3, 1, 9), (3136, 512, 1, 1, 1)])
def test_large_tile_programs_equal_the_minibatch_ones_bit_for_bit(geom, monkeypatch):
    """The 128-row forward programs and the 64 x 64 ... 32 x 256 weight-gradient programs that
    rollout- and update-sized batches select (pfrl_conv2d_nhwc_fwd / _bwd_weight), forced one after the
    other through the measurement hook (PFRL_QNET_FWD / _DGRAD / _WGRAD) on a ragged batch (B = 203:
    the last tile of every program is partial): the forward / input-gradient programs walk the
    reduction in the same order, so every output is the same bits; the small programs are the ones
    test_conv_kernels_match_torch checks against stock PyTorch."""
    C, Co, R, ST, H = geom
    dev = torch.device("cuda:0")
    B = 203
    torch.manual_seed(C)
    OH = (H - R) // ST + 1
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(Co, R, R, C, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    dy = torch.randn(B, OH, OH, Co, device=dev)
    aprev = torch.rand(B, H, H, C, device=dev) - 0.3
    lib = mt._native.lib()
    M, K = B * OH * OH, R * R * C
    nW, stride = w.numel(), w.numel() + Co
    splits = 3

    def fwd():
        y = torch.full((B, OH, OH, Co), float("nan"), device=dev)
        mt.check(lib.pfrl_conv2d_nhwc_fwd(mt._p(x), mt._p(w), mt._p(b), mt._p(y), B, H, H, C, Co, R, R, ST,
                                          1, 0, 1, mt._stream()), "fwd")
        return y

    def dgrad():
        dx = torch.full((B, H, H, C), float("nan"), device=dev)
        mt.check(lib.pfrl_conv2d_nhwc_bwd_data(mt._p(dy), None, mt._p(w), mt._p(aprev), mt._p(dx), B, H, H, C,
                                               Co, R, R, ST, 0, 0, mt._stream()), "dgrad")
        return dx

    def wgrad():
        part = torch.full((splits * stride,), float("nan"), device=dev)
        mt.check(lib.pfrl_conv2d_nhwc_bwd_weight(mt._p(dy), None, mt._p(x), mt._p(part), mt._p(part[nW:]),
                                                 stride, stride, B, H, H, C, Co, R, R, ST, splits,
                                                 mt._stream()), "wgrad")
        return part

    wide = Co % 64 == 0
    cases = [("PFRL_QNET_FWD", fwd, [3, 8] + ([2, 7] if wide else []), True)]
    if C % 32 == 0:
        # (6, 7: the stride-parity classes side by side in one tile, strided layers only)
        cases.append(("PFRL_QNET_DGRAD", dgrad, [1] + ([0] if C % 64 == 0 else [])
                      + ([6, 7, 9] if ST > 1 and 64 % C == 0 else [])
                      + ([8] if ST == 1 and R > 1 and C % 64 == 0 else []), True))     # 8: by input position
    # weight gradient: the 32 x 32 program interleaves two accumulators per tile (one MFMA tile per
    # wave), the large ones keep one: same terms, another order -> tolerance against it, and the
    # large programs bit-equal among themselves
    cases.append(("PFRL_QNET_WGRAD", wgrad, [0] + ([2] if wide and K % 64 == 0 else [])
                  + ([3] if wide and K % 128 == 0 else []) + ([4] if K % 128 == 0 else [])
                  + ([5] if K % 256 == 0 else []), False))
    for env, fn, progs, exact in cases:
        outs = []
        for prog in progs:
            monkeypatch.setenv(env, str(prog))
            outs.append(fn())
        monkeypatch.delenv(env)
        assert not torch.isnan(outs[0]).any()
        for prog, o in zip(progs[1:], outs[1:]):
            if exact:
                assert torch.equal(o, outs[0]), (env, prog)
            else:
                _close(o, outs[0], 1e-5)
                assert torch.equal(o, outs[1]), (env, prog)


@gpu
@pytest.mark.parametrize("B", [32, 7, 256])
def test_nature_trunk_forward_backward_matches_torch(B):
    dev = torch.device("cuda:0")
    ref = _nature_q()
    dut = copy.deepcopy(ref).to(dev).to(memory_format=torch.channels_last)
    mt.accelerate_heads(dut)
    torch.manual_seed(B)
    x = torch.rand(B, 4, 84, 84)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    assert mt.plan_for(dut[0].layers, dut[0].output, xg) is not None     # the kernels do run
    q_ref, q = ref(x), dut(xg)
    _close(q, q_ref, 5e-6)
    g = torch.randn_like(q_ref)
    q_ref.backward(g)
    q.backward(g.to(dev))
    for (name, p), (_, pr) in zip(dut.named_parameters(), ref.named_parameters()):
        assert p.grad.stride() == p.stride(), name
        # conv1's gradient sums 400 B terms with heavy cancellation in f32 on both sides
        _close(p.grad, pr.grad, 1e-3 if B >= 256 else 1e-5)
    # planar NCHW input is converted, not rejected
    _close(dut(x.to(dev)), q_ref, 5e-6)
    # no_grad: same values, nothing saved
    with torch.no_grad():
        assert torch.equal(dut(xg), q)


@gpu
def test_small_atari_cnn_trunk_matches_torch():
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ref = nn.Sequential(pfrl_amd.nn.SmallAtariCNN(), nn.Linear(256, 4))
    dut = copy.deepcopy(ref).to(dev).to(memory_format=torch.channels_last)
    mt.accelerate_heads(dut)
    x = torch.rand(32, 4, 84, 84)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    assert mt.plan_for(dut[0].layers, dut[0].output, xg) is not None
    q_ref, q = ref(x), dut(xg)
    _close(q, q_ref, 5e-6)
    g = torch.randn_like(q_ref)
    q_ref.backward(g)
    q.backward(g.to(dev))
    for (name, p), (_, pr) in zip(dut.named_parameters(), ref.named_parameters()):
        _close(p.grad, pr.grad, 1e-5)


@gpu
def test_sequential_trunk_fusion_on_device_matches_stock():
    """The PPO example network (examples/atari/train_ppo_ale.py:247-264) as an nn.Sequential."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    ref = nn.Sequential(nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2),
                        nn.ReLU(), nn.Conv2d(64, 64, 3), nn.ReLU(), nn.Flatten(),
                        nn.Linear(3136, 512), nn.ReLU(), nn.Linear(512, 1))
    dut = copy.deepcopy(ref).to(dev).to(memory_format=torch.channels_last)
    pfrl_amd.nn.fuse_sequential_trunk(dut)
    x = torch.rand(64, 4, 84, 84)
    v_ref = ref(x)
    v = dut(x.to(dev).contiguous(memory_format=torch.channels_last))
    _close(v, v_ref, 5e-6)
    v_ref.sum().backward()
    v.sum().backward()
    for (name, p), (_, pr) in zip(dut.named_parameters(), ref.named_parameters()):
        _close(p.grad, pr.grad, 2e-5)


@gpu
def test_trunk_in_a_captured_graph_equals_eager():
    """Forward + backward of the trunk replayed from a HIP graph (how the agents run an
    update, agents/graphed_update.py) gives the eager results bit for bit."""
    dev = torch.device("cuda:0")
    dut = _nature_q().to(dev).to(memory_format=torch.channels_last)
    mt.accelerate_heads(dut)
    xg = torch.rand(32, 4, 84, 84, device=dev).contiguous(memory_format=torch.channels_last)
    g_out = torch.randn(32, 6, device=dev)
    params = list(dut.parameters())

    def step():
        q = dut(xg)
        torch.autograd.backward([q], [g_out * 1.0])
        return q

    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(2):
            for p in params:
                p.grad = None
            step()
    cur.wait_stream(side)
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        q_graph = step()
    xg.copy_(torch.rand_like(xg))       # new input: replay, then eager on the same input
    graph.replay()
    got = [p.grad.clone() for p in params]
    q_got = q_graph.clone()
    for p in params:
        p.grad = None
    q_eager = step()
    assert torch.equal(q_got, q_eager)
    for a, p in zip(got, params):
        assert torch.equal(a, p.grad)


@gpu
def test_small_linear_head_matches_torch():
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    for M, K, N in [(32, 512, 6), (5, 512, 18 - 2), (256, 256, 1), (32, 1000, 3)]:
        head = nn.Linear(K, N)
        hg = copy.deepcopy(head).to(dev)
        h = torch.randn(M, K)
        hr = h.clone().requires_grad_(True)
        q = head(hr)
        gq = torch.randn_like(q)
        q.backward(gq)
        hx = h.to(dev).requires_grad_(True)
        assert mt.small_linear_supported(hg, hx)
        qg = mt.small_linear(hx, hg)
        qg.backward(gq.to(dev))
        _close(qg, q, 2e-6)
        _close(hx.grad, hr.grad, 2e-6)
        _close(hg.weight.grad, head.weight.grad, 5e-6)
        _close(hg.bias.grad, head.bias.grad, 5e-6)


def _device_dqn(dev, range_graphs, steps, n_envs=8, seed=0, optimizer="sgd"):
    import tempfile

    import pfrl_amd as pfrl
    from pfrl_amd import agents, explorers, replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv
    from pfrl_amd.q_functions import DiscreteActionValueHead

    pfrl.utils.set_random_seed(seed)
    store = DeviceFrameStore(4096, (84, 84), torch.uint8, dev, stack=4)
    env = SyntheticAtariVectorEnv(n_envs, store=store, seed=1, n_actions=4)
    torch.manual_seed(5)
    q = torch.nn.Sequential(pfrl.nn.LargeAtariCNN(), torch.nn.Linear(512, 4),
                            DiscreteActionValueHead()).to(memory_format=torch.channels_last)
    if optimizer == "sgd":
        opt = torch.optim.SGD(q.parameters(), lr=1e-3)
    else:
        from pfrl_amd.optimizers import FusedRMSprop

        opt = FusedRMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
    rbuf = replay_buffers.ReplayBuffer(1000)
    ex = explorers.ConstantEpsilonGreedy(0.3, lambda: np.random.randint(4))
    ag = agents.DQN(q, opt, rbuf, 0.99, ex, gpu=0, replay_start_size=64, minibatch_size=16,
                    update_interval=4, target_update_interval=48,
                    phi=lambda x: np.asarray(x, dtype=np.float32) / 255)
    ag.range_graphs = range_graphs
    if steps:
        pfrl.experiments.train_agent_batch(ag, env, steps, tempfile.mkdtemp())
    return ag, env, q


@gpu
def test_dqn_range_graph_equals_per_update_graphs():
    """All updates of an env range replayed as ONE captured graph (agents/dqn.py
    _range_as_one_graph) are the same launches in the same order as one graph per update:
    bit-identical parameters, losses and update counts, target syncs included."""
    dev = torch.device("cuda:0")
    out = []
    for rg in (True, False):
        ag, _, q = _device_dqn(dev, rg, 400)
        out.append((np.concatenate([p.detach().cpu().numpy().ravel() for p in q.parameters()]),
                    ag.loss_record.values(), ag.q_record.values(), ag.optim_t, ag.t,
                    any(k[0] == "range" for k in ag._graphed.graphs)))
    (pa, la, qa, ua, ta, ra), (pb, lb, qb, ub, tb, rb) = out
    assert ra and not rb
    assert ua == ub > 20 and ta == tb
    assert np.array_equal(la, lb) and np.array_equal(qa, qb)
    assert np.array_equal(pa, pb)


@gpu
def test_load_after_captured_updates_keeps_training_on_the_loaded_state(tmp_path):
    """optimizer.load_state_dict replaces the state tensors the captured graphs
    point at; load() must drop the graphs so that the next update steps the loaded state."""
    dev = torch.device("cuda:0")
    ag, env, q = _device_dqn(dev, True, 120, optimizer="rmsprop")
    assert ag._graphed is not None and ag._graphed.graphs
    ag.save(str(tmp_path / "ckpt"))
    saved = {k: v.clone() for k, v in q.state_dict().items()}
    import pfrl_amd as pfrl

    pfrl.experiments.train_agent_batch(ag, env, 40, str(tmp_path / "o1"), step_offset=120)
    ag.load(str(tmp_path / "ckpt"))
    assert ag._graphed is None
    for k, v in q.state_dict().items():
        assert torch.equal(v, saved[k]), k
    sq_before = [ag.optimizer.state[p]["square_avg"].clone() for p in q.parameters()]
    n0 = ag.optim_t
    pfrl.experiments.train_agent_batch(ag, env, 40, str(tmp_path / "o2"), step_offset=160)
    assert ag.optim_t > n0
    sq_after = [ag.optimizer.state[p]["square_avg"] for p in q.parameters()]
    assert any(not torch.equal(a, b) for a, b in zip(sq_before, sq_after)), \
        "the optimizer state in use is not the loaded one"


def test_graph_cache_evicts_the_least_recently_used_entry_and_keys_on_the_learning_rate():
    from pfrl_amd.agents.graphed_update import _GraphCache, _hyper_signature

    c = _GraphCache(2)
    c.admit("a", 1), c.admit("b", 2)
    assert c.lookup("a") == 1            # "a" is now the most recently used
    c.admit("c", 3)
    assert c.lookup("b") is None and c.lookup("a") == 1 and c.lookup("c") == 3
    p = torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.RMSprop([p], lr=1e-3)
    s0 = _hyper_signature([opt])
    opt.param_groups[0]["capturable"] = True      # what a capture switches on: not part of the key
    assert _hyper_signature([opt]) == s0
    opt.param_groups[0]["lr"] = 5e-4              # what a schedule hook changes: a new key
    assert _hyper_signature([opt]) != s0


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,A,clip,mean,weighted,double", [
    (32, 512, 6, True, False, False, False), (32, 512, 6, True, True, True, True),
    (48, 512, 16, False, True, False, False), (1, 256, 1, True, False, False, False),
    (100, 256, 4, True, True, True, False), (33, 512, 18 - 2, False, False, True, True),
])
def test_head_td_loss_in_one_launch_matches_the_three_launches(B, K, A, clip, mean, weighted, double):
    """pfrl_dqn_head_td_loss against: F.linear head -> the fused TD loss (itself pinned against the
    reference's compute_value_loss in test_hip_kernels) -> autograd.  Loss, y, |delta| and the
    three gradients at 1e-5 relative (same products, different summation order in the head)."""
    from pfrl_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 7 + A)
    h = torch.randn(B, K, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(A, K, generator=g) / np.sqrt(K)).to(dev).requires_grad_(True)
    b = (torch.randn(A, generator=g) * 0.1).to(dev).requires_grad_(True)
    action = torch.randint(0, A, (B,), generator=g).to(dev)
    tq = torch.randn(B, A, generator=g).to(dev)
    nq = torch.randn(B, A, generator=g).to(dev) if double else None
    r = torch.randn(B, generator=g).to(dev)
    disc = torch.full((B,), 0.99).to(dev)
    term = (torch.rand(B, generator=g) < 0.2).float().to(dev)
    wts = torch.rand(B, generator=g).to(dev) if weighted else None
    assert ops.dqn_head_td_loss_supported(h, w, b)
    loss, y, delta = ops.dqn_head_td_loss(h, w, b, action, tq, nq, r, disc, term, wts, clip, mean)
    got = torch.autograd.grad(loss, [h, w, b])
    h2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (h, w, b))
    q = torch.nn.functional.linear(h2, w2, b2)
    loss_r, y_r, delta_r = ops.dqn_td_loss(q, action, tq, nq, r, disc, term, wts, clip, mean)
    want = torch.autograd.grad(loss_r, [h2, w2, b2])
    tol = lambda ref: 1e-5 * max(ref.abs().max().item(), 1.0)
    assert abs(loss.item() - loss_r.item()) < 1e-5 * max(abs(loss_r.item()), 1.0)
    assert (y - y_r).abs().max().item() < tol(y_r) and (delta - delta_r).abs().max().item() < tol(delta_r)
    for a, ref in zip(got, want):
        assert (a - ref).abs().max().item() < tol(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [32, 1, 64])
def test_conv_only_trunk_matches_torch(N):
    """The three convolutions alone (the Rainbow head's trunk): output [N, 64, 7, 7] in NCHW
    memory against stock PyTorch fp32 on the CPU, and all gradients against the chain rule
    evaluated by torch (CPU) with the ReLU masks of the DEVICE activations: a pre-activation
    within rounding of zero may fall on either side in two summation orders, and one flipped
    mask element moves a weight gradient by a whole sample's contribution (seen at N = 32:
    1 of 165 888 elements, 2.5 % of the largest conv2 gradient entry)."""
    import copy

    from pfrl_amd.nn import mfma_trunk
    from pfrl_amd.nn.mfma_trunk import conv_fwd

    dev = torch.device("cuda:0")
    torch.manual_seed(N)
    convs = torch.nn.ModuleList([torch.nn.Conv2d(4, 32, 8, stride=4), torch.nn.Conv2d(32, 64, 4, stride=2),
                                 torch.nn.Conv2d(64, 64, 3, stride=1)]).to(dev)
    convs = convs.to(memory_format=torch.channels_last)
    x = torch.rand(N, 4, 84, 84, device=dev)
    specs = mfma_trunk.plan_for(convs, None, x)
    assert specs is not None
    out = mfma_trunk.trunk_forward(x, specs, list(convs), None)
    assert out.shape == (N, 64, 7, 7) and out.is_contiguous()
    ref = copy.deepcopy(convs).cpu()
    acts_ref = [x.cpu()]
    for c in ref:
        acts_ref.append(torch.relu(c(acts_ref[-1])))
    assert (out.cpu() - acts_ref[-1]).abs().max().item() < 1e-5 * max(acts_ref[-1].abs().max().item(), 1.0)
    # the device's own activations (NCHW views on the CPU) give the masks
    xc = x.contiguous(memory_format=torch.channels_last)
    acts_dev, h = [x.cpu()], xc
    for i, sp in enumerate(specs):
        h = conv_fwd(h, convs[i].weight, convs[i].bias, sp, N, relu=True, planar=False)
        acts_dev.append(h.permute(0, 3, 1, 2).cpu().contiguous())
    g = torch.randn(N, 3136)
    params = [p for c in convs for p in (c.weight, c.bias)]
    got = torch.autograd.grad(out.reshape(N, -1), params, g.to(dev))
    dy = g.view(N, 64, 7, 7) * (acts_dev[3] > 0)
    want = [None] * 6
    for i in (2, 1, 0):
        c = ref[i]
        a_in = acts_dev[i]
        want[2 * i] = torch.nn.grad.conv2d_weight(a_in, c.weight.shape, dy, stride=c.stride)
        want[2 * i + 1] = dy.sum((0, 2, 3))
        if i > 0:
            dy = torch.nn.grad.conv2d_input(a_in.shape, c.weight.detach().contiguous(), dy,
                                            stride=c.stride) * (a_in > 0)
    for a, r in zip(got, want):
        assert (a.cpu() - r).abs().max().item() < 2e-5 * max(r.abs().max().item(), 1.0)


# ---------------------------------------------------------------------------- u8 first layer
def test_u8_division_in_three_operations_is_ieee_division_for_every_byte():
    """The u8 operand loaders evaluate phi(x) = float32(x) / d as q = x r, q + (x - q d) r with
    r = fl(1 / d) (csrc/qnet.hip ``u8_over``).  ops.u8_division_exact decides per divisor, in
    exact rational arithmetic, whether that is IEEE division for all 256 byte values; here the
    same three operations in float64-emulated fma against NumPy's float32 division."""
    from pfrl_amd import ops

    for d in (255.0, 1.0, 256.0, 127.5):
        assert ops.u8_division_exact(d)
        d32 = np.float32(d)
        r = np.float32(1) / d32
        x = np.arange(256, dtype=np.float32)
        q = x * r
        # fma(-q, d, x) and fma(e, r, q): the products are exact in float64 (24 + 24 bits), the sums
        # of a 48-bit product and a float32 fit as well for these magnitudes
        e = (x.astype(np.float64) - q.astype(np.float64) * np.float64(d32)).astype(np.float32)
        got = (q.astype(np.float64) + e.astype(np.float64) * np.float64(r)).astype(np.float32)
        assert np.array_equal(got, x / d32), d
    assert not ops.u8_division_exact(0.0) and not ops.u8_division_exact(float("nan"))


def _u8_frames(dev, n_slots=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n_slots, 84, 84), dtype=torch.uint8, generator=g).to(dev)


@gpu
def test_raw_nhwc4_gather_holds_the_four_frames_of_each_pixel():
    from pfrl_amd import ops

    dev = torch.device("cuda:0")
    frames = _u8_frames(dev)
    refs = torch.randint(0, 300, (37, 4), dtype=torch.int32, device=dev)
    px = ops.batch_states_raw_nhwc4(frames, refs, 255.0)
    want = frames[refs.long()].permute(0, 2, 3, 1).contiguous()      # [M, H, W, 4]
    assert px.data.dtype == torch.uint8 and torch.equal(px.data, want)
    assert px.shape == (37, 4, 84, 84)
    # ... and the fp32 tensor it stands for is the one the fp32 gather writes
    assert torch.equal(px.float(), ops.batch_states_nhwc4(frames, refs, 255.0))


@gpu
@pytest.mark.parametrize("N", [16, 32, 64, 333, 512, 1001, 5300])
def test_first_convolution_on_u8_pixels_is_bit_identical_to_the_fp32_minibatch(N, monkeypatch):
    """VERDICT r4 next #4's gate: conv1 forward and weight gradient reading the u8 NHWC4
    minibatch (phi in the operand loader) == the same entries on the gathered fp32 minibatch,
    bit for bit, in every tile program the u8 entries have (forward 32 x 32 / 64 x 32 / 128 x 32,
    weight gradient 32 x 32 / 32 x 256)."""
    from pfrl_amd import _native, ops

    dev = torch.device("cuda:0")
    lib = _native.lib()
    frames = _u8_frames(dev, seed=N)
    refs = torch.randint(0, 300, (N, 4), dtype=torch.int32, device=dev)
    torch.manual_seed(N)
    conv = nn.Conv2d(4, 32, 8, stride=4).to(dev).to(memory_format=torch.channels_last)
    sp = mt.ConvSpec(conv, 84, 84)
    x32 = ops.batch_states_nhwc4(frames, refs, 255.0)
    px = ops.batch_states_raw_nhwc4(frames, refs, 255.0)
    if N >= 32:
        assert mt.u8_first_layer_ok(conv, px)
        for planar in (False, True):
            want = mt.conv_fwd(x32, conv.weight, conv.bias, sp, N, relu=True, planar=planar)
            got = mt.conv_fwd_u8(px, conv.weight, conv.bias, sp, N, relu=True, planar=planar)
            assert torch.equal(got, want), planar
        # from ~164 images both entries run the direct-form kernels (csrc/qnet.hip k_conv1_u8_direct<U8>):
        # against the tile programs they replace (PFRL_CONV1_DIRECT=0, read per call), with and without ReLU
        for relu in (True, False):
            monkeypatch.setenv("PFRL_CONV1_DIRECT", "0")
            tiles = mt.conv_fwd(x32, conv.weight, conv.bias, sp, N, relu=relu)
            tiles_u8 = mt.conv_fwd_u8(px, conv.weight, conv.bias, sp, N, relu=relu)
            monkeypatch.delenv("PFRL_CONV1_DIRECT")
            assert torch.equal(mt.conv_fwd(x32, conv.weight, conv.bias, sp, N, relu=relu), tiles)
            assert torch.equal(mt.conv_fwd_u8(px, conv.weight, conv.bias, sp, N, relu=relu), tiles_u8)
            assert torch.equal(tiles, tiles_u8)
    else:
        assert not mt.u8_first_layer_ok(conv, px)        # (under 384 tiles: the fp32 path)
    # weight gradient: partial slabs, as _Trunk._conv_backward asks for them
    M = N * sp.OH * sp.OW
    dy = torch.randn(N, sp.OH, sp.OW, sp.Cout, device=dev)
    splits = mt._wgrad_splits(M, sp.Cout, 256)
    stride = conv.weight.numel() + sp.Cout
    pa = torch.zeros(splits * stride, device=dev)
    pb = torch.zeros(splits * stride, device=dev)
    _native.check(lib.pfrl_conv2d_nhwc_bwd_weight(mt._p(dy), None, mt._p(x32), mt._p(pa),
                                                  mt._p(pa[conv.weight.numel():]), stride, stride, N,
                                                  84, 84, 4, 32, 8, 8, 4, splits, mt._stream()), "w")
    _native.check(lib.pfrl_conv2d_u8nhwc4_bwd_weight(mt._p(dy), None, mt._p(px.data), 255.0, mt._p(pb),
                                                     mt._p(pb[conv.weight.numel():]), stride, stride,
                                                     N, 84, 84, 32, 8, 8, 4, splits, mt._stream()),
                  "wu8")
    assert torch.equal(pa, pb) and float(pa.abs().max()) > 0


@gpu
def test_trunk_on_u8_pixels_equals_the_trunk_on_the_fp32_minibatch():
    """The PPO example network (examples/atari/train_ppo_ale.py:247-264) as the fused trunk:
    value and EVERY parameter gradient from a u8 minibatch (ops.U8Pixels) == from the fp32
    minibatch, bit for bit; a consumer without a u8 loader gets the fp32 tensor."""
    from pfrl_amd import ops

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = nn.Sequential(nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2),
                        nn.ReLU(), nn.Conv2d(64, 64, 3), nn.ReLU(), nn.Flatten(),
                        nn.Linear(3136, 512), nn.ReLU(), nn.Linear(512, 1))
    net = net.to(dev).to(memory_format=torch.channels_last)
    pfrl_amd.nn.fuse_sequential_trunk(net)
    frames = _u8_frames(dev, seed=9)
    refs = torch.randint(0, 300, (256, 4), dtype=torch.int32, device=dev)
    outs = []
    for u8 in (False, True):
        net.zero_grad(set_to_none=True)
        x = (ops.batch_states_raw_nhwc4(frames, refs, 255.0) if u8
             else ops.batch_states_nhwc4(frames, refs, 255.0))
        v = net(x)
        (v * torch.linspace(-1, 1, 256, device=dev)[:, None]).sum().backward()
        outs.append((v.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    # a plain (unfused) network takes the pixels as the fp32 tensor they stand for
    px = ops.batch_states_raw_nhwc4(frames, refs[:8], 255.0)
    assert torch.equal(px.float(), ops.batch_states_nhwc4(frames, refs[:8], 255.0))
