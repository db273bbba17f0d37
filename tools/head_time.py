"""Device time of pfrl_ppo_act_head at the acting (512) and value-pass (16384) batch sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pfrl_amd import ops
dev = torch.device("cuda:0")
for N in (512, 16384):
    h = torch.randn(N, 512, device=dev)
    wp, bp = torch.randn(6, 512, device=dev) * .05, torch.zeros(6, device=dev)
    wv, bv = torch.randn(1, 512, device=dev) * .05, torch.zeros(1, device=dev)
    u = torch.rand(N, device=dev)
    a = torch.randint(0, 6, (N,), device=dev)
    lp, v = torch.empty(N, device=dev), torch.empty(N, device=dev)
    for name, fn in (("sample", lambda: ops.ppo_act_head(h, wp, bp, wv, bv, u)),
                     ("given", lambda: ops.ppo_value_head(h, wp, bp, wv, bv, a, lp, v))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("N=%d %s: %.1f us" % (N, name, e0.elapsed_time(e1) * 1e3 / 20))
