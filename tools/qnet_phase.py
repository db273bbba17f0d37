"""In-kernel clocks of the forward tile program (debug build: bash tools/build_dbg.sh):
for each layer of the Nature trunk at B = 32, the span of the launch (earliest workgroup start ->
latest workgroup end, wall_clock64 at 100 MHz) against the mean lifetime of a workgroup.
    python tools/qnet_phase.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pfrl_amd import _native  # noqa: E402
DEBUG = os.environ.get("QNET_DEBUG", "1") == "1"
if DEBUG:
    _native.LIB_PATH = os.path.join(ROOT, "tools", "libpfrl_amd_dbg.so")
from pfrl_amd.nn import mfma_trunk as mt  # noqa: E402
import torch.nn as nn  # noqa: E402
L = _native.lib()
if DEBUG:
    L.pfrl_qnet_debug_read.argtypes = [ctypes.c_void_p]
    L.pfrl_qnet_debug_read.restype = ctypes.c_int
    # QREPS=2: every workgroup runs the tile program twice, the clocks are of the second pass (code
    # already fetched)
    L.pfrl_qnet_debug_set_reps(int(os.environ.get("QREPS", "1")))
else:
    class _Nop:
        def pfrl_qnet_debug_reset(self): pass
        def pfrl_qnet_debug_read(self, d): pass
    L = _Nop()
dev = torch.device("cuda:0")
B = 32
geoms = [(4, 32, 8, 4, 84), (32, 64, 4, 2, 20), (64, 64, 3, 1, 9)]
torch.manual_seed(0)
for C, Co, R, ST, H in geoms:
    conv = nn.Conv2d(C, Co, R, stride=ST).to(dev).to(memory_format=torch.channels_last)
    sp = mt.ConvSpec(conv, H, H)
    # the input is produced by another kernel right before each launch (cold in this XCD's L2)
    src = torch.rand(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    big = torch.randn(4096, 4096, device=dev)
    # the same inside a captured graph (20 x [producer, conv]): total workgroup lifetime / count,
    # and the graph's wall time per pair
    import time
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            x = (src + 1.0).contiguous(memory_format=torch.channels_last)
            mt.conv_fwd(x, conv.weight, conv.bias, sp, B, relu=True, planar=False)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(g):
        for _ in range(20):
            x = (src + 1.0).contiguous(memory_format=torch.channels_last)
            y = mt.conv_fwd(x, conv.weight, conv.bias, sp, B, relu=True, planar=False)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(20):
            x = (src + 1.0).contiguous(memory_format=torch.channels_last)
    for gg in (g, g2):
        gg.replay()
    torch.cuda.synchronize()
    L.pfrl_qnet_debug_reset()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t_pair = (time.perf_counter() - t0) / 400 * 1e6
    t0 = time.perf_counter()
    for _ in range(20):
        g2.replay()
    torch.cuda.synchronize()
    t_prod = (time.perf_counter() - t0) / 400 * 1e6
    print("conv %dx%d/%d C%d->%d in a graph: [producer + conv] %.2f us, producer alone %.2f us -> conv %.2f us"
          % (R, R, ST, C, Co, t_pair, t_prod, t_pair - t_prod))
    if DEBUG:
        import numpy as np
        buf = np.zeros((4096, 8), dtype=np.uint64)
        L.pfrl_qnet_debug_read(buf.ctypes.data)
        st_ = buf[buf[:, 5] > 0].astype(np.int64)       # stamps of the LAST launch of the graph
        t0 = st_[:, 0].min()
        us = lambda x: x / 100.0                        # noqa: E731  (100 MHz ticks)
        print("   %d workgroups; start spread %.2f us (first -> last workgroup entry), launch span %.2f us"
              % (len(st_), us(st_[:, 0].max() - t0), us(st_[:, 5].max() - t0)))
        names = ["entry -> first-stage loads issued", "-> first stage parked in LDS (data landed)",
                 "-> pipeline done", "-> accumulators folded", "-> end (epilogue stores issued)"]
        for k, nm in enumerate(names):
            d = st_[:, k + 1] - st_[:, k]
            print("     %-46s mean %.2f  p10 %.2f  p90 %.2f us" % (nm, us(d.mean()), us(np.percentile(d, 10)),
                                                                  us(np.percentile(d, 90))))
        d = st_[:, 6] - st_[:, 2]
        print("     [first loop stage] next stage's loads issued       mean %.2f us" % us(d.mean()))
        d = st_[:, 7] - st_[:, 6]
        print("     [first loop stage] G chunks computed out of LDS      mean %.2f us" % us(d.mean()))
        life = st_[:, 5] - st_[:, 0]
        print("     workgroup lifetime                             mean %.2f  max %.2f us" % (us(life.mean()), us(life.max())))
