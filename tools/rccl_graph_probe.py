"""Can this stack capture an RCCL all-reduce inside a HIP graph?  (1 rank)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("eager ok", float(x[0]))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = x * 2
        dist.all_reduce(y, op=dist.ReduceOp.AVG)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        dist.all_reduce(y, op=dist.ReduceOp.AVG)
        z = y + 1
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("captured ok", float(z[0]))
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print("replay us", (time.perf_counter() - t0) / 200 * 1e6)
    t0 = time.perf_counter()
    for _ in range(200):
        y = x * 2
        dist.all_reduce(y, op=dist.ReduceOp.AVG)
        z = y + 1
    torch.cuda.synchronize()
    print("eager us", (time.perf_counter() - t0) / 200 * 1e6)
except Exception as e:
    print("capture FAILED:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
