"""Per-kernel cost inside torch HIP graphs: homogeneous vs mixed chains, tiny vs real kernels."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfrl_amd.nn import mfma_trunk as mt
import torch.nn as nn

dev = torch.device("cuda:0")

def gtime(fn, per, reps=100):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / per * 1e6

x = torch.zeros(256, device=dev); y = torch.zeros(32, 6, device=dev)
def e1():
    for _ in range(64): x.add_(1)
print("E1 64x add_                 %.2f us/kernel" % gtime(e1, 64))
def e2():
    for _ in range(16):
        y.fill_(1.0); x.add_(1); y.mul_(2.0); y.sum()
print("E2 fill/add/mul/sum x16     %.2f us/kernel" % gtime(e2, 64))
part = torch.randn(14, 32 * 512, device=dev); out = torch.empty(32 * 512, device=dev); b = torch.randn(512, device=dev)
def e3():
    for _ in range(32): mt._reduce([(part, out, None, 32 * 512, 1024, 1, 4, 0)])
print("E3 reduce tiny (1 split)    %.2f us/kernel" % gtime(e3, 32))
def e4():
    for _ in range(32): mt._reduce([(part, out, b, 32 * 512, 32 * 512, 14, 512, 1)])
print("E4 reduce 14 splits x 16K   %.2f us/kernel" % gtime(e4, 32))
conv2 = nn.Conv2d(32, 64, 4, stride=2).to(dev).to(memory_format=torch.channels_last)
x2 = torch.rand(32, 20, 20, 32, device=dev)
sp2 = mt.ConvSpec(conv2, 20, 20)
def e5():
    for _ in range(32): mt.conv_fwd(x2, conv2.weight, conv2.bias, sp2, 32)
print("E5 conv2 fwd B=32           %.2f us/kernel" % gtime(e5, 32))
lin = nn.Linear(32, 32).to(dev); xl = torch.rand(32, 32, device=dev)
def e6():
    for _ in range(32): mt.linear_fwd(xl, lin.weight, lin.bias)
print("E6 linear 32x32x32 (1 chunk) %.2f us/kernel" % gtime(e6, 32))
conv1 = nn.Conv2d(4, 32, 8, stride=4).to(dev).to(memory_format=torch.channels_last)
x1 = torch.rand(32, 84, 84, 4, device=dev); sp1 = mt.ConvSpec(conv1, 84, 84)
conv3 = nn.Conv2d(64, 64, 3, stride=1).to(dev).to(memory_format=torch.channels_last)
x3 = torch.rand(32, 9, 9, 64, device=dev); sp3 = mt.ConvSpec(conv3, 9, 9)
def e7():
    for _ in range(16):
        mt.conv_fwd(x1, conv1.weight, conv1.bias, sp1, 32); mt.conv_fwd(x3, conv3.weight, conv3.bias, sp3, 32)
print("E7 conv1/conv3 fwd alternating %.2f us/kernel" % gtime(e7, 32))
def e8():
    for _ in range(32): mt.conv_fwd(x1, conv1.weight, conv1.bias, sp1, 32)
print("E8 conv1 fwd B=32           %.2f us/kernel" % gtime(e8, 32))
lin1 = nn.Linear(3136, 512).to(dev); xf = torch.rand(32, 3136, device=dev)
def e9():
    for _ in range(16): mt.linear_fwd(xf, lin1.weight, lin1.bias)
print("E9 fc1 fwd (split + reduce) %.2f us/pair" % gtime(e9, 16))
