"""Time stock F.linear (hipBLASLt) against the MFMA linear kernels at MLP-agent shapes,
inside captured graphs (the way the update loop runs them).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pfrl_amd.nn import mfma_trunk as mt

dev = torch.device("cuda:0")

def graph_time(fn, reps=50, inner=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for K, N in ((376, 256), (393, 256), (256, 256), (384, 256), (256, 34), (256, 1), (512, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    t_f = graph_time(lambda: F.linear(x, w, b))
    t_dx = graph_time(lambda: dy @ w)
    t_dw = graph_time(lambda: dy.t() @ x)
    line = "M=%d K=%d N=%d  stock fwd %.1f us  dx %.1f us  dw %.1f us" % (M, K, N, t_f, t_dx, t_dw)
    if K % 32 == 0 and N % 32 == 0:
        t_m = graph_time(lambda: mt.linear_fwd(x, w, b, relu=True))
        ref = torch.relu(F.linear(x, w, b)); got = mt.linear_fwd(x, w, b, relu=True)
        line += "  | mfma fwd %.1f us (maxerr %.2e)" % (t_m, (ref - got).abs().max().item())
    print(line, flush=True)

# whole layers through the autograd node (forward + backward), both split rules
from pfrl_amd.nn import mfma_linear as ml
import torch.nn as nn
for direct in (96, 10 ** 9):
    ml._DIRECT_TILES = direct
    for K, N, relu in ((376, 256, True), (393, 256, True), (256, 256, True), (256, 34, False)):
        lin = nn.Linear(K, N).to(dev); slot = ml._LinearSlot(lin)
        x = torch.randn(M, K, device=dev, requires_grad=True); dy = torch.randn(M, N, device=dev)
        def fb():
            y = slot(x, relu=relu)
            return torch.autograd.grad(y, [x, slot.weight, slot.bias], dy)
        def fb_ref():
            y = F.linear(x, lin.weight, lin.bias)
            if relu: y = torch.relu(y)
            return torch.autograd.grad(y, [x, lin.weight, lin.bias], dy)
        with torch.no_grad():
            t_f = graph_time(lambda: slot(x, relu=relu))
        print("direct_tiles=%d M=%d K=%d N=%d: node fwd %.1f us, fwd+bwd %.1f us (stock fwd+bwd %.1f us)"
              % (direct, M, K, N, t_f, graph_time(fb), graph_time(fb_ref)), flush=True)
