"""In-kernel phase timing of k_tree_sample_lds (debug build).

Build the instrumented library first (from the repo root):
    for f in frames replay sumtree rollout optim; do hipcc --offload-arch=gfx950 -O3 -std=c++17 \
        -fPIC -ffp-contract=off -DPFRL_TREE_DEBUG -c pfrl_amd/csrc/$f.hip -o /tmp/$f.o; done
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/{frames,replay,sumtree,rollout,optim}.o \
        -o tools/libpfrl_amd_dbg.so
Phases per draw: top-heap descent, subtree fan-out load, bottom descent, repair, write-back.
Round-1 result at capacity 1e6 (us/draw): 1.9 / 2.0 / 1.4 / 1.6 / 0.2 = 7.1 (global-memory
descent of the first version: 11.3).
"""
import ctypes, os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pfrl_amd import _native
_native.LIB_PATH = os.path.join(root, "tools", "libpfrl_amd_dbg.so")
from pfrl_amd.collections.prioritized import PrioritizedBuffer
cap = 10 ** 6
dev = torch.device("cuda:0")
buf = PrioritizedBuffer(cap, device=dev)
rs = np.random.RandomState(0)
for i in range(cap + 5000):
    buf.append(i)
    if (i & 1023) == 1023:
        buf.flush()
buf.flush()
L = _native.lib()
L.pfrl_tree_debug_read.argtypes = [ctypes.c_void_p]
# keep the GPU busy/clocked with a background matmul stream
a = torch.randn(4096, 4096, device=dev)
for B in (32, 32, 32):
    for _ in range(20):
        a @ a
    out = buf.sample_device(B, u01=rs.random_sample(B))
    torch.cuda.synchronize()
    dbg = (ctypes.c_ulonglong * 8)()
    L.pfrl_tree_debug_read(dbg)
    t = np.array(list(dbg)[:5], dtype=np.float64) / B
    print("per draw (wall_clock64 ticks @100MHz -> us):", (t / 100.0).round(2), "sum", (t.sum() / 100).round(2))
    buf.set_last_priority([1.0] * B)
