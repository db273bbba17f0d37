"""Where one replay-agent update spends its time ON THE DEVICE, without a profiler in the way:
CUDA events recorded around the graph replays (main stream) and around the tree / gather launches
(replay stream) of the bench workload; prints the mean offsets of every mark from the start of the
update's forward graph.   python tools/pipeline_events.py --algo rainbow [--updates 64]"""
import collections, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
n_upd = 64
if "--updates" in sys.argv:
    i = sys.argv.index("--updates"); n_upd = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
cap = 200000
if "--capacity" in sys.argv:
    i = sys.argv.index("--capacity"); cap = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
sys.argv = ["bench.py", "--no-cpu-baseline", "--capacity", str(cap)] + sys.argv[1:]
args = bench.parse_args()
dev = torch.device("cuda:0")
agent, env, rbuf = bench.build_agent(args, dev, 0)
N = args.num_envs
obss = env.reset()
obss = bench.prefill(agent, env, obss, N, (cap if cap > 200000 else 60000) if args.algo != "sac" else 20000)
for _ in range(6):
    obss = bench.one_step(agent, env, obss, N)
torch.cuda.synchronize()

from pfrl_amd import ops  # noqa: E402
marks = []          # (name, event)
on = [False]

def mark(name):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(dev))
        marks.append((name, e))

def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        mark(name + ":begin")
        r = f(*a, **k)
        mark(name + ":end")
        return r
    setattr(obj, attr, g)

_replay = torch.cuda.CUDAGraph.replay
_count = [0]
_names = {}
def replay(self):
    # graphs named in the order they are first replayed inside the measured region (an update
    # replays two or three: [noise,] forward, backward + step)
    name = _names.setdefault(id(self), "graph%d" % len(_names)) if on[0] else "graph?"
    mark(name + ":begin")
    r = _replay(self)
    mark(name + ":end")
    _count[0] += 1
    return r
torch.cuda.CUDAGraph.replay = replay
for attr in ("tree_sample", "batch_experiences", "tree_update_errors_write_f32", "tree_update_errors_f32",
             "tree_write", "table_append", "entries_append"):
    if hasattr(ops, attr):
        wrap(ops, attr, attr)
# modules that imported the functions by name keep their own references: patch those too
import pfrl_amd.collections.prioritized as cp  # noqa: E402
import pfrl_amd.replay_buffers.device_replay as dr  # noqa: E402
_count[0] = 0
on[0] = True
t0 = time.perf_counter()
steps = max(1, n_upd // 64)
for _ in range(steps):
    obss = bench.one_step(agent, env, obss, N)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
on[0] = False
# split into updates at graph0:begin
idx = [i for i, (n, _) in enumerate(marks) if n == "graph0:begin"]
rows = collections.defaultdict(list)
period = []
for a, b in zip(idx[:-1], idx[1:]):
    base = marks[a][1]
    period.append(base.elapsed_time(marks[b][1]) * 1e3)
    # marks that belong to this update: from the previous graph1:end (exclusive) ... the tree work
    # launched for THIS update precedes graph0:begin on the host, so look back to the previous graph0
    lo = idx[idx.index(a) - 1] if idx.index(a) > 0 else 0
    seen = collections.Counter()
    for n, e in marks[lo + 1:b]:
        seen[n] += 1
        rows["%s#%d" % (n, seen[n])].append(base.elapsed_time(e) * 1e3)
print("updates %d, wall per update %.1f us, device period (graph0 begin to next) mean %.1f us"
      % (len(period), wall / max(1, len(idx)) * 1e6, float(np.mean(period))))
order = sorted(rows.items(), key=lambda kv: np.mean(kv[1]))
for n, v in order:
    if len(v) >= len(period) // 2:
        print("%10.1f us  (sd %6.1f, n %3d)  %s" % (np.mean(v), np.std(v), len(v), n))
