"""In-kernel phase timing of the PER sampler (debug build, tools/build_dbg.sh) on a tree in the
state the Rainbow bench keeps it in: Python-float leaves (max_priority appends) with np.float32
leaves where minibatches were drawn, np.float32 sums above them.
    PFRL_TREE_SAMPLE=paths|lds python tools/per_dbg2.py [capacity]
Phases (us per draw): paths: top rounds / fan-out / bottom round + siblings / repair / write-back
                      lds:   top descent / fan-out / bottom descent / repair / write-back"""
import ctypes, os, sys, time, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pfrl_amd import _native
_native.LIB_PATH = os.path.join(root, "tools", os.environ.get("PER_DBG_LIB", "libpfrl_amd_dbg.so"))
from pfrl_amd.collections.prioritized import PrioritizedBuffer
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
dev = torch.device("cuda:0")
buf = PrioritizedBuffer(cap, device=dev)
rs = np.random.RandomState(0)
t0 = time.perf_counter()
for i in range(cap + 5000):
    buf.append(i)
    if (i & 1023) == 1023:
        buf.flush()
buf.flush()
torch.cuda.synchronize()
print("fill %.1f s" % (time.perf_counter() - t0), flush=True)
for _ in range(100):
    out = buf.sample_device(1024, u01=rs.random_sample(1024))
    err = torch.from_numpy((rs.rand(1024) * 1.2).astype(np.float32)).to(dev)
    buf.update_errors_device(err, 0, 0.01 ** 0.5, 1, 1.01 ** 0.5, 0.01, 0.5)
buf.flush()
torch.cuda.synchronize()
L = _native.lib()
L.pfrl_tree_debug_read.argtypes = [ctypes.c_void_p]
a = torch.randn(4096, 4096, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (32, 32, 32, 32):
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    if os.environ.get("PER_DBG_LOAD"):
        # a streaming kernel beside the sampler, as the optimizer step is in an update
        if "big" not in globals():
            big = torch.zeros(1 << 28, device=dev)      # 1 GiB
            side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(12):
                big.mul_(1.0001)
    u = rs.random_sample(B)
    ev0.record()
    out = buf.sample_device(B, u01=u, normalize=2, beta=0.5, slot_mod=cap)
    ev1.record()
    torch.cuda.synchronize()
    dbg = (ctypes.c_ulonglong * 8)()
    L.pfrl_tree_debug_read(dbg)
    t = np.array(list(dbg)[:5], dtype=np.float64) / (B if os.environ.get("PFRL_TREE_SAMPLE", "prefetch") != "prefetch" else B * B / B)
    print(os.environ.get("PFRL_TREE_SAMPLE", "prefetch"), "per draw us (prefetch: whole-launch prologue / draws / epilogue):", (t / 100.0).round(2), "sum", (t.sum() / 100).round(2),
          "| launch(es) by events %.1f us" % (ev0.elapsed_time(ev1) * 1e3), flush=True)
    err = torch.from_numpy((rs.rand(B) * 1.2).astype(np.float32)).to(dev)
    buf.update_errors_device(err, 0, 0.01 ** 0.5, 1, 1.01 ** 0.5, 0.01, 0.5)
    buf.flush()
