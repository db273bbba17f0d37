"""Actor-critic update helpers (csrc/actor.hip) against the PyTorch code they replace.

GPU tests call through the C ABI (pfrl_squashed_gaussian_fwd/_bwd, pfrl_soft_update,
pfrl_adam_step, pfrl_linear_small_bwd) and compare with torch.distributions / torch.optim /
the reference's soft-update formula evaluated by torch on the same device.  Tolerances: the
soft update is bit-exact (three rounded f32 operations in the reference's order); Adam and the
distribution are the same f32 formulas with at most a different association inside libm calls
and reductions: 1e-6 relative / 2e-5 on the summed log-probability.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
from torch import distributions as D

from pfrl_amd.optimizers import FusedAdam
from pfrl_amd.utils.copy_param import soft_copy_param, soft_copy_params
from pfrl_amd.utils.squashed_gaussian import sample_with_log_prob, squashed_gaussian_params


def _head(x, cache_size=1):
    mean, log_scale = torch.chunk(x, 2, dim=1)
    scale = torch.sqrt(torch.exp(torch.clamp(log_scale, -20.0, 2.0) * 2))
    return D.TransformedDistribution(D.Independent(D.Normal(loc=mean, scale=scale), 1),
                                     [D.transforms.TanhTransform(cache_size=cache_size)])


def test_cpu_distributions_take_their_own_methods():
    torch.manual_seed(0)
    x = torch.randn(6, 8)
    d = _head(x)
    assert squashed_gaussian_params(d) is None      # CPU tensors
    torch.manual_seed(1)
    a, lp = sample_with_log_prob(d, True)
    torch.manual_seed(1)
    a2 = d.rsample()
    assert torch.equal(a, a2) and torch.allclose(lp, d.log_prob(a2))


def test_soft_update_cpu_is_the_reference_formula():
    torch.manual_seed(0)
    src, dst = nn.Linear(5, 3), nn.Linear(5, 3)
    want = {k: (1 - 0.01) * v + 0.01 * src.state_dict()[k] for k, v in dst.state_dict().items()}
    soft_copy_param(dst, src, 0.01)
    for k, v in dst.state_dict().items():
        assert torch.allclose(v, want[k], rtol=0, atol=1e-7)


def test_fused_adam_on_cpu_is_torch_adam():
    torch.manual_seed(0)
    a, b = nn.Linear(4, 4), None
    b = copy.deepcopy(a)
    oa, ob = FusedAdam(a.parameters(), lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    for _ in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(torch.ones(2, 4)).sum().backward()
            o.step()
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    assert isinstance(oa, torch.optim.Adam)


@pytest.mark.parametrize("with_closure", [False, True])
def test_fused_rmsprop_off_the_gpu_is_torch_rmsprop_once_per_call(with_closure):
    """Parameters the kernel does not cover (here: CPU tensors, two groups) take torch's step --
    exactly one step per call for every group, also when a closure is passed (the fused class
    used to return the closure's loss without stepping at all)."""
    from pfrl_amd.optimizers import FusedRMSprop

    torch.manual_seed(0)
    a = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 2))
    b = copy.deepcopy(a)

    def groups(m):
        return [{"params": m[0].parameters()}, {"params": m[1].parameters(), "lr": 3e-3}]

    oa = FusedRMSprop(groups(a), lr=1e-2, alpha=0.95, eps=1e-2, centered=True)
    ob = torch.optim.RMSprop(groups(b), lr=1e-2, alpha=0.95, eps=1e-2, centered=True)
    for _ in range(3):
        for m, o in ((a, oa), (b, ob)):
            def closure(m=m, o=o):
                o.zero_grad()
                loss = m(torch.ones(2, 4)).sum()
                loss.backward()
                return loss
            if with_closure:
                assert o.step(closure) is not None
            else:
                closure()
                o.step()
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(256, 17), (32, 6), (1, 1), (100, 70), (7, 3)])
def test_squashed_gaussian_matches_torch_distributions(B, A):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 100 + A)
    raw = (torch.randn(B, 2 * A, generator=g) * 1.5).to(dev)
    g_a = torch.randn(B, A, generator=g).to(dev)
    g_lp = torch.randn(B, generator=g).to(dev)
    outs = []
    for fused in (True, False):
        x = raw.clone().requires_grad_(True)
        d = _head(x)
        assert (squashed_gaussian_params(d) is not None) == True
        torch.manual_seed(5)
        if fused:
            a, lp = sample_with_log_prob(d, True)
        else:
            a = d.rsample()
            lp = d.log_prob(a)
        (gx,) = torch.autograd.grad([a, lp], [x], [g_a, g_lp])
        outs.append((a.detach(), lp.detach(), gx))
    (a1, lp1, gx1), (a0, lp0, gx0) = outs
    assert torch.allclose(a1, a0, rtol=1e-6, atol=1e-7)
    assert torch.allclose(lp1, lp0, rtol=2e-5, atol=2e-5 * A)
    scale = max(gx0.abs().max().item(), 1.0)
    assert (gx1 - gx0).abs().max().item() < 2e-5 * scale
    # sampling without reparameterisation: same draw, no graph
    torch.manual_seed(5)
    a2, lp2 = sample_with_log_prob(_head(raw), False)
    assert torch.equal(a2, a1) and torch.equal(lp2, lp1) and not a2.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(256, 17), (32, 6), (1, 1), (100, 70)])
def test_fused_policy_head_equals_the_head_function_plus_the_distribution_launches(B, A):
    """pfrl_squashed_head_fwd/_bwd (the example's head folded in) against the user's function +
    pfrl_squashed_gaussian_fwd/_bwd + autograd through chunk / clamp / exp / sqrt: same draw,
    same sample, same log-probability, same input gradient -- log-scales on both sides of both
    clamp bounds included (the mask of clamp's backward)."""
    from pfrl_amd.utils.squashed_gaussian import head_sample_with_log_prob, recognise_head

    dev = torch.device("cuda:0")
    spec = recognise_head(_head, 2 * A, dev)
    assert spec is not None and (spec.lo, spec.hi, spec.mode, spec.A) == (-20.0, 2.0, 0, A)
    g = torch.Generator().manual_seed(B * 100 + A)
    raw = torch.randn(B, 2 * A, generator=g) * 1.5
    raw[:, A:] = raw[:, A:] * 8.0 - 6.0            # log-scales from about -30 to 18
    raw = raw.to(dev)
    g_a = torch.randn(B, A, generator=g).to(dev)
    g_lp = torch.randn(B, generator=g).to(dev)
    outs = []
    for fused in (True, False):
        x = raw.clone().requires_grad_(True)
        torch.manual_seed(5)
        if fused:
            a, lp, neg = head_sample_with_log_prob(x, spec, True)
            assert torch.equal(neg, -lp.detach())
        else:
            a, lp = sample_with_log_prob(_head(x), True)
        (gx,) = torch.autograd.grad([a, lp], [x], [g_a, g_lp])
        outs.append((a.detach(), lp.detach(), gx))
    (a1, lp1, gx1), (a0, lp0, gx0) = outs
    assert torch.allclose(a1, a0, rtol=1e-6, atol=1e-7)
    assert torch.allclose(lp1, lp0, rtol=1e-6, atol=1e-5)
    assert torch.equal(gx1[:, A:] == 0, gx0[:, A:] == 0)       # the clamp mask, element for element
    scale = max(gx0.abs().max().item(), 1.0)
    assert (gx1 - gx0).abs().max().item() <= 1e-6 * scale
    if spec.bit_exact:
        assert torch.equal(a1, a0) and torch.equal(lp1, lp0) and torch.equal(gx1, gx0)
    torch.manual_seed(5)
    a2, lp2, _ = head_sample_with_log_prob(raw, spec, False)
    assert torch.equal(a2, a1) and torch.equal(lp2, lp1) and not a2.requires_grad


@pytest.mark.gpu
def test_policy_head_recognition_takes_only_what_it_proved():
    from pfrl_amd.utils.squashed_gaussian import recognise_head

    dev = torch.device("cuda:0")

    def squashed(loc, scale):
        return D.TransformedDistribution(D.Independent(D.Normal(loc=loc, scale=scale), 1),
                                         [D.transforms.TanhTransform(cache_size=1)])

    def exp_head(x):
        mean, ls = torch.chunk(x, 2, dim=1)
        return squashed(mean, torch.exp(torch.clamp(ls, -5.0, 1.5)))

    spec = recognise_head(exp_head, 12, dev)
    assert spec is not None and (spec.lo, spec.hi, spec.mode, spec.A) == (-5.0, 1.5, 1, 6)

    def softplus_head(x):
        mean, ls = torch.chunk(x, 2, dim=1)
        return squashed(mean, torch.nn.functional.softplus(ls) + 1e-3)

    def swapped_head(x):
        ls, mean = torch.chunk(x, 2, dim=1)
        return squashed(mean, torch.exp(torch.clamp(ls, -20.0, 2.0)))

    def scaled_mean_head(x):
        mean, ls = torch.chunk(x, 2, dim=1)
        return squashed(mean * 0.5, torch.exp(torch.clamp(ls, -20.0, 2.0)))

    def tanh_bounded_head(x):      # (log-scale squashed into the range instead of clamped)
        mean, ls = torch.chunk(x, 2, dim=1)
        return squashed(mean, torch.exp(-20.0 + 11.0 * (torch.tanh(ls) + 1.0)))

    for fn in (softplus_head, swapped_head, scaled_mean_head, tanh_bounded_head,
               lambda x: _head(x, cache_size=0), lambda x: x):
        assert recognise_head(fn, 12, dev) is None
    assert recognise_head(_head, 13, dev) is None          # odd width: no (mean | log_scale) halves
    state = torch.cuda.get_rng_state(dev)
    recognise_head(_head, 12, dev)
    assert torch.equal(torch.cuda.get_rng_state(dev), state)   # the probe draws from its own generator


@pytest.mark.gpu
def test_other_distributions_are_left_alone():
    dev = torch.device("cuda:0")
    raw = torch.randn(4, 6, device=dev)
    assert squashed_gaussian_params(_head(raw, cache_size=0)) is None
    n = D.Independent(D.Normal(raw[:, :3], torch.ones(4, 3, device=dev)), 1)
    assert squashed_gaussian_params(n) is None
    a, lp = sample_with_log_prob(n, True)
    assert torch.allclose(lp, n.log_prob(a))


@pytest.mark.gpu
def test_soft_update_is_bit_exact_and_one_launch_covers_many_tensors():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mk = lambda: nn.Sequential(nn.Linear(393, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                               nn.Linear(256, 1), nn.BatchNorm1d(1)).to(dev)
    srcs = [mk() for _ in range(5)]      # 5 x 8 float tensors: more than one kernel-argument block
    dsts = [mk() for _ in range(5)]
    for s in srcs:
        s[5].num_batches_tracked += 3
    tau = 5e-3
    want = []
    for s, d in zip(srcs, dsts):
        want.append({k: ((1 - tau) * v + tau * s.state_dict()[k]) if v.is_floating_point()
                     else s.state_dict()[k].clone() for k, v in d.state_dict().items()})
    soft_copy_params(list(zip(dsts, srcs)), tau)
    for d, w in zip(dsts, want):
        for k, v in d.state_dict().items():
            assert torch.equal(v, w[k]), k


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam_and_keeps_its_state_layout():
    """Same gradients on both sides (Adam turns last-bit gradient differences of near-zero
    entries into lr-sized update differences, so the gradients are not recomputed per device):
    the device step against torch's single-tensor Adam on the CPU."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(376, 256), nn.ReLU(), nn.Linear(256, 34))
    ref = copy.deepcopy(net)                       # stays on the CPU
    net.to(dev)
    g = torch.Generator().manual_seed(1)
    for wd in (0.0, 1e-2):
        o = FusedAdam(net.parameters(), lr=3e-4, weight_decay=wd)
        r = torch.optim.Adam(ref.parameters(), lr=3e-4, weight_decay=wd)
        for step in range(6):
            for p, q in zip(net.parameters(), ref.parameters()):
                q.grad = torch.randn(q.shape, generator=g) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=g)))
                p.grad = q.grad.to(dev)
            o.step()
            r.step()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert (p.cpu() - q).abs().max().item() <= 2e-7 * max(q.abs().max().item(), 1e-3)
            st = o.state[p]
            assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 6.0
            for key in ("exp_avg", "exp_avg_sq"):
                ref_v = r.state[q][key]
                assert (st[key].cpu() - ref_v).abs().max().item() <= 1e-6 * ref_v.abs().max().item()
        # the state dict loads into a stock Adam and back
        stock = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=wd, capturable=True)
        stock.load_state_dict(o.state_dict())
        o.load_state_dict(stock.state_dict())
        o.step()
        assert float(o.state[next(net.parameters())]["step"]) == 7.0
        for p, q in zip(net.parameters(), ref.parameters()):
            q.data.copy_(p.detach().cpu())         # next round starts from equal parameters


@pytest.mark.gpu
def test_fused_adam_step_in_a_captured_graph():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    p = nn.Parameter(torch.randn(5000, device=dev))
    q = nn.Parameter(p.detach().clone())
    o, r = FusedAdam([p], lr=1e-2), torch.optim.Adam([q], lr=1e-2, capturable=True)
    grad = torch.randn(5000, device=dev)
    p.grad, q.grad = grad.clone(), grad.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        o.step()
        r.step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o.step()
    for _ in range(4):
        graph.replay()
        r.step()
    torch.cuda.synchronize()
    assert float(o.state[p]["step"]) == 5.0 + 0.0   # 1 eager + 4 replays (capture does not run)
    assert (p - q).abs().max().item() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(256, 256, 1), (32, 512, 6), (100, 300, 1), (5, 40, 16), (256, 256, 16)])
def test_narrow_head_backward_matches_torch(M, K, N):
    from pfrl_amd.nn.mfma_trunk import _SmallLinear

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True)
    dy = torch.randn(M, N, generator=g).to(dev)
    y = _SmallLinear.apply(x, w, b)
    got = torch.autograd.grad(y, [x, w, b], dy)
    y_ref = torch.nn.functional.linear(x, w, b)
    want = torch.autograd.grad(y_ref, [x, w, b], dy)
    assert torch.allclose(y, y_ref, rtol=1e-5, atol=1e-5)
    for a, r in zip(got, want):
        assert (a - r).abs().max().item() < 2e-5 * max(r.abs().max().item(), 1.0)


def test_sac_losses_on_cpu_are_the_reference_formulas():
    from pfrl_amd.agents import _sac_losses as L

    torch.manual_seed(0)
    B = 9
    r, d, t = torch.randn(B), torch.rand(B), (torch.rand(B) < 0.3).float()
    q1, q2, lp = torch.randn(B, 1), torch.randn(B, 1), torch.randn(B)
    want = r + d * (1.0 - t) * torch.flatten(torch.min(q1, q2) - 0.2 * lp[..., None])
    assert torch.allclose(L.soft_target_q(r, d, t, q1, q2, lp, 0.2), want)
    log_t = torch.tensor(-0.7)
    want = r + d * (1.0 - t) * torch.flatten(torch.min(q1, q2) - torch.exp(log_t) * lp[..., None])
    assert torch.allclose(L.soft_target_q(r, d, t, q1, q2, lp, log_t), want)
    p = torch.randn(B, requires_grad=True)
    assert torch.allclose(L.half_mse(r, p), 0.5 * torch.nn.functional.mse_loss(r, p))
    assert torch.allclose(L.policy_loss(lp, q1, q2, 0.2), torch.mean(0.2 * lp[..., None] - torch.min(q1, q2)))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [256, 1, 100, 1000])
def test_sac_loss_kernels_match_the_torch_formulas(B):
    from pfrl_amd.agents import _sac_losses as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    r, d = rnd(B), torch.rand(B, generator=g).to(dev)
    t = (torch.rand(B, generator=g) < 0.3).float().to(dev)
    q1, q2, lp = rnd(B, 1), rnd(B, 1), rnd(B)
    q2[::7] = q1[::7]                                   # ties in the minimum
    log_t = torch.nn.Parameter(torch.tensor(-0.7, device=dev))
    for temp, tval in ((0.2, 0.2), (log_t, torch.exp(log_t.detach()))):
        want = r + d * (1.0 - t) * torch.flatten(torch.min(q1, q2) - tval * lp[..., None])
        got = L.soft_target_q(r, d, t, q1, q2, lp, temp)
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
        # policy loss and its three gradients
        a, b, c = (x.clone().requires_grad_(True) for x in (lp, q1, q2))
        loss = L.policy_loss(a, b, c, temp)
        ga = torch.autograd.grad(loss, [a, b, c])
        a2, b2, c2 = (x.clone().requires_grad_(True) for x in (lp, q1, q2))
        ref = torch.mean(tval * a2[..., None] - torch.min(b2, c2))
        gr = torch.autograd.grad(ref, [a2, b2, c2])
        assert torch.allclose(loss, ref, rtol=1e-5, atol=1e-6)
        for x, y in zip(ga, gr):
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-8)
    p = rnd(B).requires_grad_(True)
    loss = L.half_mse(r, p)
    (gp,) = torch.autograd.grad(loss, [p])
    p2 = p.detach().clone().requires_grad_(True)
    ref = 0.5 * torch.nn.functional.mse_loss(r, p2)
    (gr,) = torch.autograd.grad(ref, [p2])
    assert torch.allclose(loss, ref, rtol=1e-5) and torch.allclose(gp, gr, rtol=1e-5, atol=1e-8)
    # temperature loss: value and d/dlog T
    from pfrl_amd.agents.soft_actor_critic import TemperatureHolder

    holder = TemperatureHolder(-0.7).to(dev)
    loss = L.temperature_loss(holder, lp, -17.0)
    (gt,) = torch.autograd.grad(loss, [holder.log_temperature])
    ref = -torch.mean(holder() * (lp + (-17.0)))
    (gr,) = torch.autograd.grad(ref, [holder.log_temperature])
    assert torch.allclose(loss, ref, rtol=1e-5) and torch.allclose(gt, gr, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [256, 100, 1])
def test_loss_gradients_written_by_the_forward_launch_equal_the_backward_kernels(B):
    """``loss.backward(unit_grad(device))`` (what the SAC agent passes): the gradients the forward
    launch wrote for dL/dloss = 1 (unit_g_* of pfrl_half_mse_twin_fwd / pfrl_sac_policy_loss_fwd)
    are bit for bit what the _bwd kernels compute from any other tensor holding 1."""
    from pfrl_amd.agents import _sac_losses as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + 11)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    target, lp, q1, q2 = rnd(B), rnd(B), rnd(B, 1), rnd(B, 1)
    q2[::5] = q1[::5]
    log_t = torch.nn.Parameter(torch.tensor(-0.3, device=dev))
    unit, other = L.unit_grad(dev), torch.ones((), device=dev)
    grads = []
    for one in (unit, other):
        a, b, c = (x.clone().requires_grad_(True) for x in (lp, q1, q2))
        gl = torch.autograd.grad(L.policy_loss(a, b, c, log_t), [a, b, c], [one])
        p1, p2 = (x.clone().requires_grad_(True) for x in (q1.flatten(), q2.flatten()))
        l1, l2 = L.half_mse_pair(target, p1, p2)
        gm = torch.autograd.grad([l1, l2], [p1, p2], [one, one])
        grads.append(list(gl) + list(gm))
    for x, y in zip(*grads):
        assert x.shape == y.shape and torch.equal(x, y)
    # one unit and one foreign upstream gradient: the backward kernels
    p1, p2 = (x.clone().requires_grad_(True) for x in (q1.flatten(), q2.flatten()))
    l1, l2 = L.half_mse_pair(target, p1, p2)
    gm = torch.autograd.grad([l1, l2], [p1, p2], [unit, other * 2.0])
    assert torch.equal(gm[0], grads[0][3]) and torch.allclose(gm[1], grads[0][4] * 2.0, rtol=1e-6)


@pytest.mark.gpu
def test_fused_adam_step_together_equals_separate_steps():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    nets = [nn.Linear(393, 256).to(dev) for _ in range(2)]
    refs = [copy.deepcopy(n) for n in nets]
    opts = [FusedAdam(n.parameters(), lr=3e-4) for n in nets]
    ropts = [FusedAdam(n.parameters(), lr=3e-4) for n in refs]
    for _ in range(3):
        for n, r in zip(nets, refs):
            for p, q in zip(n.parameters(), r.parameters()):
                p.grad = torch.randn_like(p)
                q.grad = p.grad.clone()
        FusedAdam.step_together(opts)
        for o in ropts:
            o.step()
    for n, r, o in zip(nets, refs, opts):
        for p, q in zip(n.parameters(), r.parameters()):
            assert torch.equal(p, q) and float(o.state[p]["step"]) == 3.0
    # different hyperparameters, or a stock optimizer in the list: one after the other
    mixed = [opts[0], torch.optim.Adam(nets[1].parameters(), lr=1e-3, capturable=True)]
    FusedAdam.step_together(mixed)
    assert float(opts[0].state[next(nets[0].parameters())]["step"]) == 4.0


def test_cpu_routes_of_the_round_two_helpers():
    """Off the GPU every helper takes PyTorch's own route: several optimizers step one after the
    other, several (target, source) pairs soft-update tensor by tensor, and the twin / MFMA
    layer plans decline."""
    from pfrl_amd.nn import accelerate_mlp
    from pfrl_amd.nn.twin_mlp import twin_forward

    import pfrl_amd as pfrl

    torch.manual_seed(0)
    nets = [nn.Linear(6, 4) for _ in range(2)]
    refs = [copy.deepcopy(n) for n in nets]
    opts = [FusedAdam(n.parameters(), lr=1e-2) for n in nets]
    ropts = [torch.optim.Adam(n.parameters(), lr=1e-2) for n in refs]
    for n, r in zip(nets, refs):
        for p, q in zip(n.parameters(), r.parameters()):
            p.grad = torch.randn_like(p)
            q.grad = p.grad.clone()
    FusedAdam.step_together(opts)
    for o in ropts:
        o.step()
    for n, r in zip(nets, refs):
        assert all(torch.equal(p, q) for p, q in zip(n.parameters(), r.parameters()))
    # soft update of two pairs at once == one pair after the other
    src, dst = [nn.Linear(5, 3) for _ in range(2)], [nn.Linear(5, 3) for _ in range(2)]
    want = [copy.deepcopy(d) for d in dst]
    for s, w in zip(src, want):
        soft_copy_param(w, s, 0.05)
    soft_copy_params(list(zip(dst, src)), 0.05)
    for d, w in zip(dst, want):
        assert all(torch.equal(a, b) for a, b in zip(d.state_dict().values(), w.state_dict().values()))
    # twin plan: not on the CPU
    q = lambda: accelerate_mlp(nn.Sequential(pfrl.nn.ConcatObsAndAction(), nn.Linear(11, 32), nn.ReLU(),
                                             nn.Linear(32, 32), nn.ReLU(), nn.Linear(32, 1)))
    q1, q2 = q(), q()
    assert twin_forward(q1, q2, (torch.randn(4, 8), torch.randn(4, 3))) is None


@pytest.mark.gpu
def test_fused_adam_sums_gradient_slabs_and_soft_updates_the_targets_in_its_launch():
    """pfrl_adam_step_ex against the three launches it replaces -- pfrl_splitk_reduce (fold the
    split-K slabs), pfrl_adam_step, pfrl_soft_update -- on the layers of the SAC critics: same
    parameters, moments and targets, bit for bit, over several steps."""
    from pfrl_amd.nn import mfma_trunk as _t

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    nets = [nn.Linear(393, 256).to(dev), nn.Linear(256, 256).to(dev)]
    refs = copy.deepcopy(nets)
    tgt, tgt_ref = copy.deepcopy(nets), copy.deepcopy(nets)
    opts = [FusedAdam(n.parameters(), lr=3e-4) for n in nets]
    opts_ref = [FusedAdam(n.parameters(), lr=3e-4) for n in refs]
    tau = 5e-3
    for it in range(4):
        slabs, soft = {}, {}
        for net, ref, t_net, splits in zip(nets, refs, tgt, (5 + it, 11)):
            nW, Fo = net.weight.numel(), net.bias.numel()
            stride = nW + Fo
            part = torch.randn(splits * stride, device=dev)
            slabs[net.weight.data_ptr()] = (part, stride, splits)
            slabs[net.bias.data_ptr()] = (part[nW:], stride, splits)
            soft[net.weight.data_ptr()], soft[net.bias.data_ptr()] = t_net.weight.data, t_net.bias.data
            dw, db = torch.empty_like(ref.weight), torch.empty_like(ref.bias)
            _t._reduce([(part, dw, None, stride, nW, splits, 4, 0), (part[nW:], db, None, stride, Fo, splits, 4, 0)])
            ref.weight.grad, ref.bias.grad = dw, db
            net.weight.grad = net.bias.grad = None
        assert FusedAdam.step_together(opts, slabs=slabs, soft=soft, tau=tau) is True
        FusedAdam.step_together(opts_ref)
        soft_copy_params(list(zip(tgt_ref, refs)), tau)
        for a, b in zip(nets + tgt, refs + tgt_ref):
            assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
        for o, orf, n, r in zip(opts, opts_ref, nets, refs):
            for p, q in zip(n.parameters(), r.parameters()):
                assert torch.equal(o.state[p]["exp_avg"], orf.state[q]["exp_avg"])
                assert torch.equal(o.state[p]["exp_avg_sq"], orf.state[q]["exp_avg_sq"])
                assert float(o.state[p]["step"]) == float(orf.state[q]["step"]) == it + 1
    # a single optimizer's step() with slabs only
    net, ref = nn.Linear(64, 32).to(dev), None
    ref = copy.deepcopy(net)
    o, orf = FusedAdam(net.parameters(), lr=1e-3), FusedAdam(ref.parameters(), lr=1e-3)
    part = torch.randn(3 * (64 * 32 + 32), device=dev)
    view = part.view(3, -1)
    ref.weight.grad = (view[0, :2048] + view[1, :2048] + view[2, :2048]).view(32, 64)
    ref.bias.grad = view[0, 2048:] + view[1, 2048:] + view[2, 2048:]
    o.step(slabs={net.weight.data_ptr(): (part, 2080, 3), net.bias.data_ptr(): (part[2048:], 2080, 3)})
    orf.step()
    assert torch.equal(net.weight, ref.weight) and torch.equal(net.bias, ref.bias)


@pytest.mark.gpu
def test_sac_update_with_riders_equals_the_separate_launches(monkeypatch):
    """The SAC update of the bench model (FusedAdam, 256-256 networks, B = 256) with the gradient
    slabs summed in the optimizer launches and the soft target update riding in the critics' step
    (PFRL_SAC_RIDERS, default on) against the same update with the fold, step and soft-update
    launches separate: every network and target network bit for bit after 12 updates."""
    import pfrl_amd as pfrl
    from pfrl_amd import agents, replay_buffers
    from pfrl_amd.envs.synthetic import HostSyntheticVectorObsEnv
    from pfrl_amd.nn import ConcatObsAndAction, Lambda

    def run(riders):
        monkeypatch.setenv("PFRL_SAC_RIDERS", riders)
        obs_dim, act_dim, N = 376, 17, 4
        pfrl.utils.set_random_seed(0)
        torch.manual_seed(11)
        env = HostSyntheticVectorObsEnv(N, obs_dim=obs_dim, act_dim=act_dim, seed=2, p_done=0.02)
        policy = nn.Sequential(nn.Linear(obs_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                               nn.Linear(256, act_dim * 2), Lambda(_head))

        def q():
            return nn.Sequential(ConcatObsAndAction(), nn.Linear(obs_dim + act_dim, 256), nn.ReLU(),
                                 nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1))

        q1, q2 = q(), q()
        opts = [FusedAdam(m.parameters(), lr=3e-4) for m in (policy, q1, q2)]
        ag = agents.SoftActorCritic(
            policy, q1, q2, opts[0], opts[1], opts[2], replay_buffers.ReplayBuffer(2000), gamma=0.99,
            gpu=0, replay_start_size=300, minibatch_size=256, update_interval=4,
            burnin_action_func=lambda: np.random.uniform(-1, 1, size=act_dim).astype(np.float32),
            entropy_target=-act_dim, temperature_optimizer_lr=3e-4)
        obs = env.reset()
        for _ in range(88):
            acts = ag.batch_act(obs)
            obs, r, done, _ = env.step(acts)
            ag.batch_observe(obs, r, done, np.zeros(N, dtype=bool))
            obs = env.reset(np.logical_not(done))
        assert ag.n_policy_updates >= 12
        mods = (ag.policy, ag.q_func1, ag.q_func2, ag.target_q_func1, ag.target_q_func2,
                ag.temperature_holder)
        return [p.detach().clone() for m in mods for p in m.parameters()], ag

    on, ag = run("1")
    assert ag._soft_done is True
    off, ag0 = run("0")
    assert ag0._soft_done is False
    assert len(on) == len(off)
    for a, b in zip(on, off):
        assert torch.equal(a, b)
