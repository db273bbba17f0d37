"""N > 1 supervision, the stall watchdog, ``also`` workloads in their own processes (bench.py)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BENCH = os.path.join(ROOT, "bench.py")

class _StallWatchdog:
    """N > 1 only: if the ranks stop making progress (a peer died, a collective hangs), rank 0
    still prints a line -- ``value`` null, ``config.dp_plan`` = "fallback:stalled ..." -- and the
    process leaves with status 0 instead of sitting in the driver's timeout."""

    def __init__(self, args, rank, world, result_fd, limit_s):
        import threading

        self.t = time.time()
        self.what = "start"
        self.done = False
        self.printed = False
        self.args, self.rank, self.world, self.fd, self.limit = args, rank, world, result_fd, limit_s
        threading.Thread(target=self._run, name="pfrl-bench-watchdog", daemon=True).start()

    def tick(self, what):
        self.t, self.what = time.time(), what

    def _run(self):
        while not self.done:
            time.sleep(1.0)
            if time.time() - self.t > self.limit:
                a = self.args
                line = {"metric": "env-steps/sec whole node (%s)" % a.algo.upper(), "value": None,
                        "unit": "env-steps/s", "n_gpus": self.world, "steps": a.steps,
                        "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                        "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                        "config": {"workload": "not completed", "ranks_seen": self.world,
                                   "dp_plan": "fallback:stalled for %.0f s in %s" % (self.limit, self.what)}}
                if self.rank == 0 and not self.printed:
                    os.write(self.fd, (json.dumps(line) + "\n").encode())
                os._exit(0)


_WATCHDOG = [None]


def _tick(what):
    if _WATCHDOG[0] is not None:
        _WATCHDOG[0].tick(what)


def _die_with_parent():
    import ctypes
    import signal

    ctypes.CDLL("libc.so.6").prctl(1, signal.SIGKILL)       # PR_SET_PDEATHSIG


DP_PLANS = [
    ("captured collectives (RCCL inside the update graph)", {}),
    ("eager RCCL collective between two graphs", {"PFRL_GRAPH_COLLECTIVE": "0"}),
    ("process group, host-staged", {"PFRL_RCCL_DIRECT": "0", "PFRL_GRAPH_COLLECTIVE": "0"}),
]


def supervise(args):
    """N > 1: every rank launched by torchrun is a SUPERVISOR that never touches the GPU; the
    workload runs in a child process (this same file, PFRL_BENCH_CHILD=1, its own rendezvous
    port).  A child that dies (a SIGSEGV inside hipStreamEndCapture with RCCL nodes in the graph is
    not an exception anyone can catch -- round 5 met exactly that with a live peer), stalls or
    prints no value costs ONE attempt: the supervisors tell each other through the rendezvous store,
    stop their children, and start the next, more conservative plan of DP_PLANS.  Rank 0 prints the
    first line that every rank completed, with ``config.dp_attempts`` listing what failed before;
    if nothing completes it prints a line with ``value`` null.  Exit status 0 either way: the data
    plane never costs the driver its JSON line."""
    import subprocess

    import torch.distributed as dist

    # (gloo's connection banner goes to fd 1 through C stdio: park fd 1 on stderr, as main() does)
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    base_port = int(os.environ.setdefault("MASTER_PORT", "29500"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    limit = float(os.environ.get("PFRL_BENCH_ATTEMPT_S", "600"))
    attempts, line = [], None
    first = int(os.environ.get("PFRL_BENCH_FIRST_PLAN", "0"))
    for k, (name, extra) in list(enumerate(DP_PLANS))[first:]:
        env = dict(os.environ)
        env.update(extra)
        env.update(PFRL_BENCH_CHILD="1", MASTER_PORT=str(base_port + 101 + k))
        for v in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID"):
            env.pop(v, None)       # (the children rendezvous on a store of their own)
        argv = [sys.executable, BENCH] + sys.argv[1:]
        if os.environ.get("PFRL_BENCH_CHILD_ARGV"):        # (tests/test_bench_supervisor.py: a stub worker)
            argv = json.loads(os.environ["PFRL_BENCH_CHILD_ARGV"])
        child = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, preexec_fn=_die_with_parent)
        key = "pfrl_bench_attempt_%d_failed" % k
        t0, why, out = time.time(), None, b""
        while True:
            try:
                out, _ = child.communicate(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                if store.add(key, 0) > 0:
                    why = "stopped: a peer's worker failed"
                elif time.time() - t0 > limit:
                    why = "no result within %.0f s" % limit
                if why is not None:
                    child.kill()
                    out, _ = child.communicate()
                    break
        parsed = None
        if why is None and child.returncode != 0:
            why = "worker exited with status %d" % child.returncode
        if why is None and rank == 0:
            try:
                parsed = json.loads(out.decode().strip().splitlines()[-1])
                if parsed.get("value") is None:
                    why = "worker printed no value (%s)" % parsed.get("config", {}).get("dp_plan")
            except Exception as e:      # noqa: BLE001
                why = "worker printed no JSON line (%s)" % (e,)
        if why is not None:
            store.add(key, 1)
        ok = torch.tensor([0.0 if why is not None else 1.0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0.5:
            line = parsed
            break
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        attempts.append({"plan": name, "failed": {str(r): w for r, w in enumerate(reasons) if w}})
        sys.stderr.write("bench.py supervisor: plan '%s' failed (%s)\n" % (name, attempts[-1]["failed"]))
    if rank == 0:
        if line is None:
            line = {"metric": "env-steps/sec whole node (%s)" % args.algo.upper(), "value": None,
                    "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling,
                    "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "not completed", "ranks_seen": world,
                               "dp_plan": "fallback:every data-parallel plan failed"}}
        line.setdefault("config", {})["dp_attempts_failed"] = attempts
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    os.close(result_fd)
    dist.barrier()
    dist.destroy_process_group()


def also_in_own_process(args, argv, limit_s=900, extra_env=None):
    """One ``also`` workload as ``bench.py --algo X`` would measure it ALONE: a process of its own,
    so that nothing an earlier workload of this process left behind (allocator state, captured
    graphs and their pools, module-level hooks of another agent) is part of the number.  (Round 5:
    Rainbow measured 5.8 k env-steps/s as the fifth workload of one process against 9.3-9.9 k alone
    or straight after PPO; the line is about each workload, not about their order.)  Returns the
    child's result dict, or None (the caller then runs the workload in this process)."""
    import subprocess

    cmd = [sys.executable, BENCH] + argv + [
        "--no-also", "--no-cpu-baseline", "--no-data-path-only", "--seed", str(args.seed)]
    if args.allow_lib_override:
        cmd.append("--allow-lib-override")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PFRL_BENCH_CHILD"):
        env.pop(k, None)
    env.update(extra_env or {})
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env,
                           timeout=limit_s)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stderr.write("bench.py: %s exited with %d; running it in this process\n"
                             % (" ".join(argv), r.returncode))
            return None
        d = json.loads(lines[-1])
        d["config"]["process"] = "its own: " + " ".join(
            ["%s=%s" % kv for kv in sorted((extra_env or {}).items())] + ["bench.py"] + argv)
        return d
    except Exception as e:      # (timeout, unparsable line: the workload still gets measured)
        sys.stderr.write("bench.py: %s in its own process failed (%s); running it in this process\n"
                         % (" ".join(argv), e))
        return None
