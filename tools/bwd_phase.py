"""In-kernel clocks of the fused backward launches (input gradient + weight gradient of one layer in
one grid) of the Nature trunk at B = 32 -- debug build (bash tools/build_dbg.sh).  For each launch:
how many workgroups run each tile program, their mean lifetime, and when the last one of each kind
ends relative to the first workgroup's entry.
    python tools/bwd_phase.py            (GRIDS=1764,684,848: workgroups of the launches to look at)"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from pfrl_amd import _native  # noqa: E402
_native.LIB_PATH = os.path.join(ROOT, "tools", "libpfrl_amd_dbg.so")
import qnet_check as q  # noqa: E402
L = _native.lib()
L.pfrl_qnet_debug_read.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
ref, dut = q.make_model(dev)
B = int(os.environ.get("B", "32"))
xg = torch.rand(B, 4, 84, 84, device=dev).contiguous(memory_format=torch.channels_last)


def fb():
    for p in dut.parameters():
        p.grad = None
    dut(xg).sum().backward()


st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for _ in range(3):
        fb()
torch.cuda.current_stream().wait_stream(st)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fb()
names = {1764: "hidden layer 3136 -> 512", 684: "conv3 3x3/1 64 -> 64", 848: "conv2 4x4/2 32 -> 64"}
for grid in [int(v) for v in os.environ.get("GRIDS", "1764,684,848").split(",")]:
    L.pfrl_qnet_debug_set_grid(grid)
    L.pfrl_qnet_debug_reset()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    L.pfrl_qnet_debug_read(buf.ctypes.data)
    s = buf[buf[:, 5] > 0].astype(np.int64)
    if not len(s):
        print("no launch with %d workgroups" % grid)
        continue
    t0 = s[:, 0].min()
    print("%s: %d workgroups recorded, launch span %.2f us" % (names.get(grid, grid), len(s), (s[:, 5].max() - t0) / 100.0))
    for kind, nm in ((1, "input gradient "), (2, "weight gradient")):
        k = s[s[:, 4] == kind]
        if not len(k):
            continue
        life = (k[:, 5] - k[:, 0]) / 100.0
        print("   %s %5d workgroups  lifetime mean %.2f p90 %.2f us   first entry +%.2f  last entry +%.2f  last end +%.2f us"
              % (nm, len(k), life.mean(), np.percentile(life, 90), (k[:, 0].min() - t0) / 100.0,
                 (k[:, 0].max() - t0) / 100.0, (k[:, 5].max() - t0) / 100.0))
        print("        entry -> first loads issued %.2f, -> first stage parked %.2f, -> end %.2f us (means)"
              % ((k[:, 1] - k[:, 0]).mean() / 100.0, (k[:, 2] - k[:, 1]).mean() / 100.0, (k[:, 5] - k[:, 2]).mean() / 100.0))
