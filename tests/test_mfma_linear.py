"""nn.Linear (+ ReLU) of the MLP agents on the MFMA kernels (pfrl_amd/nn/mfma_linear.py).

CPU: `accelerate_mlp` leaves parameters, names, state_dict keys and CPU results untouched.
GPU: forward and all three gradients against torch's own F.linear / relu in f32, through the
C ABI (`pfrl_linear_fwd`, `pfrl_conv2d_nhwc_bwd*`), at the shapes of the SAC / TD3 example
networks (in_features 376 and 376 + 17: not multiples of 32, rows not 16-byte aligned).
Tolerance 2e-5 relative to the output scale: same f32 products, different summation order.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from pfrl_amd.nn import accelerate_mlp
from pfrl_amd.nn import mfma_linear as ml


def _sac_q(obs=376, act=17):
    import pfrl_amd as pfrl

    return nn.Sequential(pfrl.nn.ConcatObsAndAction(), nn.Linear(obs + act, 256), nn.ReLU(),
                         nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1))


def test_accelerate_mlp_keeps_parameters_names_and_cpu_results():
    torch.manual_seed(0)
    q = _sac_q()
    ref = copy.deepcopy(q)
    keys = list(q.state_dict().keys())
    params = [p for p in q.parameters()]
    accelerate_mlp(q)
    assert list(q.state_dict().keys()) == keys
    assert all(a is b for a, b in zip(q.parameters(), params))
    assert isinstance(q, nn.Sequential) and isinstance(q[1], nn.Linear) and isinstance(q[2], nn.ReLU)
    x = (torch.randn(5, 376), torch.randn(5, 17))
    assert torch.equal(q(x), ref(x))
    # a deep copy (the agents' target networks) keeps working and shares nothing
    t = copy.deepcopy(q)
    assert torch.equal(t(x), ref(x))
    assert t[1].weight is not q[1].weight
    # load_state_dict from an unaccelerated network
    q.load_state_dict(ref.state_dict())


def test_split_rule():
    assert ml._fwd_splits(256, 256, 256) == 1
    s = ml._fwd_splits(32, 256, 376)
    assert 1 <= s <= 12 and ml._fwd_splits(32, 34, 256) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,relu", [
    (256, 376, 256, True), (256, 393, 256, True), (256, 256, 256, True), (256, 256, 34, False),
    (100, 400, 300, True), (100, 300, 300, True), (7, 33, 40, True), (1, 376, 256, True),
    (64, 376, 256, True), (32, 256, 256, False), (1000, 64, 64, True), (5, 3, 17, False),
    (100, 64, 62, False), (33, 100, 50, False), (256, 256, 12 + 17, False),
])
def test_linear_matches_torch(M, K, N, relu):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 1000 + K + N)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    lin = nn.Linear(K, N)
    lin.weight.data = torch.randn(N, K, generator=g) / np.sqrt(K)
    lin.bias.data = torch.randn(N, generator=g) * 0.1
    lin = lin.to(dev)
    slot = ml._LinearSlot(copy.deepcopy(lin))
    dy = torch.randn(M, N, generator=g).to(dev)
    y_ref = torch.nn.functional.linear(x, lin.weight, lin.bias)
    if relu:
        y_ref = torch.relu(y_ref)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, [x, lin.weight, lin.bias], dy)
    assert ml.supported(slot, x)
    y = slot(x, relu=relu)
    gx, gw, gb = torch.autograd.grad(y, [x, slot.weight, slot.bias], dy)
    for name, a, b in (("y", y, y_ref), ("dx", gx, gx_ref), ("dw", gw, gw_ref), ("db", gb, gb_ref)):
        scale = max(b.abs().max().item(), 1e-6)
        err = (a - b).abs().max().item() / scale
        assert err < 2e-5, (name, err)
    if relu:
        assert torch.equal(y > 0, y_ref > 0) or ((y > 0) != (y_ref > 0)).float().mean().item() < 1e-3


@pytest.mark.gpu
def test_accelerated_sac_networks_match_the_stock_route_and_survive_a_graph():
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    q = _sac_q().to(dev)
    ref = copy.deepcopy(q)
    accelerate_mlp(q)
    assert isinstance(q, ml._MlpSequential) and isinstance(q[1], ml._LinearSlot)
    obs, act = torch.randn(256, 376, device=dev), torch.randn(256, 17, device=dev, requires_grad=True)
    out, out_ref = q((obs, act)), ref((obs, act))
    assert (out - out_ref).abs().max().item() < 2e-5 * max(out_ref.abs().max().item(), 1.0)
    out.sum().backward()
    ga = act.grad.clone()
    act.grad = None
    out_ref.sum().backward()
    assert (ga - act.grad).abs().max().item() < 2e-5 * max(act.grad.abs().max().item(), 1.0)
    for p, r in zip(q.parameters(), ref.parameters()):
        assert (p.grad - r.grad).abs().max().item() < 2e-5 * max(r.grad.abs().max().item(), 1.0)
    # inside a captured graph: replays give what the eager run gave
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(2):
            q((obs, act))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        captured = q((obs, act))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, out.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("K,N", [(256, 256), (393, 256), (256, 1), (256, 34)])
def test_frozen_weights_give_the_input_gradient_only(K, N):
    """requires_grad off on the layer (the Q-networks under the SAC policy loss): the input
    gradient is unchanged and no weight gradient is produced."""
    from pfrl_amd.nn import accelerate_mlp

    dev = torch.device("cuda:0")
    torch.manual_seed(K + N)
    seq = nn.Sequential(nn.Linear(K, N), nn.ReLU()).to(dev) if N > 16 else nn.Sequential(nn.Linear(K, N)).to(dev)
    ref = copy.deepcopy(seq)
    accelerate_mlp(seq)
    x = torch.randn(256, K, device=dev, requires_grad=True)
    dy = torch.randn(256, N, device=dev)
    for m in (seq, ref):
        m.requires_grad_(False)
    (gx,) = torch.autograd.grad(seq(x), [x], dy)
    (gr,) = torch.autograd.grad(ref(x), [x], dy)
    assert (gx - gr).abs().max().item() < 2e-5 * max(gr.abs().max().item(), 1.0)
    seq.requires_grad_(True)
    y = seq(x)
    y.backward(dy)
    assert all(p.grad is not None for p in seq.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("M,obs,act,hid", [(256, 376, 17, 256), (100, 24, 3, 64), (7, 11, 33, 32),
                                           (33, 20, 6, 32), (48, 40, 32, 96)])
def test_twin_q_networks_match_the_two_modules(M, obs, act, hid):
    """twin_forward(q1, q2, (s, a)) against q1((s, a)), q2((s, a)) evaluated one by one by
    stock PyTorch: values, all parameter gradients, the action gradient; then with frozen
    parameters (the policy loss): the action gradient only."""
    from pfrl_amd.nn import accelerate_mlp
    from pfrl_amd.nn.twin_mlp import twin_forward, twin_plan

    dev = torch.device("cuda:0")
    torch.manual_seed(M + obs)

    def mk():
        import pfrl_amd as pfrl

        return nn.Sequential(pfrl.nn.ConcatObsAndAction(), nn.Linear(obs + act, hid), nn.ReLU(),
                             nn.Linear(hid, hid), nn.ReLU(), nn.Linear(hid, 1)).to(dev)

    q1, q2 = mk(), mk()
    r1, r2 = copy.deepcopy(q1), copy.deepcopy(q2)
    accelerate_mlp(q1), accelerate_mlp(q2)
    s = torch.randn(M, obs, device=dev)
    a = torch.randn(M, act, device=dev, requires_grad=True)
    a_ref = a.detach().clone().requires_grad_(True)
    g1, g2 = torch.randn(M, 1, device=dev), torch.randn(M, 1, device=dev)
    assert twin_plan(q1, q2, (s, a)) is not None
    o1, o2 = twin_forward(q1, q2, (s, a))
    w1, w2 = r1((s, a_ref)), r2((s, a_ref))
    tol = lambda ref: 2e-5 * max(ref.abs().max().item(), 1.0)
    assert (o1 - w1).abs().max().item() < tol(w1) and (o2 - w2).abs().max().item() < tol(w2)
    torch.autograd.backward([o1, o2], [g1, g2])
    torch.autograd.backward([w1, w2], [g1, g2])
    assert (a.grad - a_ref.grad).abs().max().item() < tol(a_ref.grad)
    for q, r in ((q1, r1), (q2, r2)):
        for p, pr in zip(q.parameters(), r.parameters()):
            assert (p.grad - pr.grad).abs().max().item() < tol(pr.grad)
    # one network's loss only (the other output unused): its gradients only
    for q in (q1, q2):
        q.zero_grad()
    o1, o2 = twin_forward(q1, q2, (s, a.detach()))
    o1.backward(g1)
    for p, pr in zip(q1.parameters(), r1.parameters()):
        assert (p.grad - pr.grad).abs().max().item() < tol(pr.grad)
    assert all(float(p.grad.abs().max()) == 0.0 for p in q2.parameters())
    # frozen parameters
    a.grad = None
    for q in (q1, q2):
        q.zero_grad()
        q.requires_grad_(False)
    o1, o2 = twin_forward(q1, q2, (s, a))
    torch.autograd.backward([o1, o2], [g1, g2])
    assert (a.grad - a_ref.grad).abs().max().item() < tol(a_ref.grad)
    assert all(p.grad is None for q in (q1, q2) for p in q.parameters())
    # mixed requires_grad, a CPU input, a different shape: not twinned
    q1.requires_grad_(True)
    assert twin_forward(q1, q2, (s, a)) is None
    assert twin_forward(q1, q1, (s.cpu(), a.detach().cpu())) is None
