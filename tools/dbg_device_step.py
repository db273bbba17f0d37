import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from test_bench_path_parity import _bench_args
dev = torch.device("cuda:0")
N = 256
args = _bench_args(capacity=1500, frame_slots=6144, slack=512, replay_start=1024)
agent, env, rbuf = bench.build_agent(args, dev, 0)
from pfrl_amd import ops
from pfrl_amd.agents import _dqn_device_step as dds
if os.environ.get("VARIANT") == "A":
    dds.plan_range = lambda *a: None
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        r = fn(*a, **k)
        torch.cuda.synchronize(); print("   ", name, "ok", flush=True)
        return r
    setattr(mod, name, w)
for n in ("select_actions", "frames_synth_u8", "frames_synth_u8_ring", "batch_states_nhwc4", "batch_states", "batch_experiences", "table_append", "entries_append"):
    wrap(ops, n)
obss = env.reset()
torch.cuda.synchronize(); print("reset ok", flush=True)
for step in range(12):
    a = agent.batch_act(obss)
    torch.cuda.synchronize(); print(step, "act ok", type(a).__name__, flush=True)
    obss2, rs, dones, _ = env.step(a)
    torch.cuda.synchronize(); print(step, "env ok", flush=True)
    agent.batch_observe(obss2, rs, dones, np.zeros(N, dtype=bool))
    torch.cuda.synchronize(); print(step, "observe ok", len(rbuf), flush=True)
    st = rbuf.store
    for nm in ("state_ref", "next_ref", "reward", "terminal"):
        d = getattr(st, "t_" + nm).cpu().numpy(); h = getattr(st, "h_" + nm)
        if not np.array_equal(d, h): print("   MISMATCH", nm, np.flatnonzero((d != h).reshape(len(d), -1).any(1))[:10], flush=True)
    et = st.e_tids.cpu().numpy()[:, 0]; hl = st.h_e_len
    live = np.arange(rbuf.memory.head, st.n_entries) % st.E
    want = (st.h_e_tids[live, 0] % st.R)
    if not np.array_equal(et[live], want): print("   MISMATCH e_tids", flush=True)
    if not np.array_equal(st.e_len.cpu().numpy()[live], st.h_e_len[live]): print("   MISMATCH e_len", flush=True)
    print("   tables ok; frames", hex(st.frames.frames.data_ptr()), st.frames.frames.numel(), "n_trans", st.n_trans, "head", rbuf.memory.head, flush=True)
    print(step, "dones", np.flatnonzero(dones), flush=True)
    obss = env.reset(np.logical_not(dones))
    torch.cuda.synchronize(); print(step, "reset ok", flush=True)
print("done")
