"""bench.py --gpus N (N > 1): the per-rank supervisor that keeps the driver's JSON line alive when a
data-parallel plan takes a worker process down (SIGSEGV inside a graph capture with RCCL nodes, a
hang in a collective).  CPU only: the workers are stubs, the supervisors are the real code."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


STUB = textwrap.dedent('''
    import json, os, signal, sys, time
    rank = int(os.environ["RANK"])
    assert os.environ["PFRL_BENCH_CHILD"] == "1" and "TORCHELASTIC_USE_AGENT_STORE" not in os.environ
    mode = os.environ["STUB_MODE"]
    captured = os.environ.get("PFRL_GRAPH_COLLECTIVE") != "0"
    direct = os.environ.get("PFRL_RCCL_DIRECT") != "0"
    if mode == "segv_then_ok" and captured:
        if rank == 1:
            os.kill(os.getpid(), signal.SIGSEGV)      # dies in "hipStreamEndCapture"
        time.sleep(120)                                # its peer waits in a collective
    if mode == "always_fail" and rank == 0:
        sys.exit(3)
    if mode == "null_until_process_group" and direct:
        if rank == 0:
            print(json.dumps({"value": None, "config": {"dp_plan": "fallback:stalled"}}))
        sys.exit(0)
    if rank == 0:
        print("noise on stdout")
        print(json.dumps({"value": 123.0, "config": {"port": os.environ["MASTER_PORT"],
                                                    "captured": captured, "direct": direct}}))
''')


def _run(tmp_path, mode, timeout=90):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), STUB_MODE=mode,
                   TORCHELASTIC_USE_AGENT_STORE="False", PFRL_BENCH_ATTEMPT_S="30",
                   PFRL_BENCH_CHILD_ARGV=json.dumps([sys.executable, str(stub)]))
        env.pop("PFRL_BENCH_CHILD", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert outs[1][0].strip() == b""                 # only rank 0 prints
    lines = outs[0][0].decode().strip().splitlines()
    assert len(lines) == 1                            # exactly ONE line
    return json.loads(lines[0]), port


def test_a_worker_that_segfaults_costs_one_attempt_not_the_line(tmp_path):
    line, port = _run(tmp_path, "segv_then_ok")
    assert line["value"] == 123.0 and line["config"]["captured"] is False and line["config"]["direct"]
    failed = line["config"]["dp_attempts_failed"]
    assert len(failed) == 1 and "captured" in failed[0]["plan"]
    assert "status -11" in failed[0]["failed"]["1"]            # the SIGSEGV
    assert "peer" in failed[0]["failed"]["0"]                  # its peer was stopped, not waited for
    assert int(line["config"]["port"]) == port + 102            # the second plan's own rendezvous


def test_a_worker_without_a_value_moves_on_to_the_next_plan(tmp_path):
    line, _ = _run(tmp_path, "null_until_process_group")
    assert line["value"] == 123.0 and line["config"]["direct"] is False
    assert [a["plan"].split()[0] for a in line["config"]["dp_attempts_failed"]] == ["captured", "eager"]


def test_when_every_plan_fails_the_line_is_still_printed_with_status_zero(tmp_path):
    line, _ = _run(tmp_path, "always_fail")
    assert line["value"] is None and line["n_gpus"] == 2
    assert len(line["config"]["dp_attempts_failed"]) == 3
    assert line["config"]["dp_plan"].startswith("fallback:")


def test_also_workload_in_its_own_process_falls_back_when_the_child_fails(monkeypatch):
    """bench.py measures every ``also`` workload as ``bench.py --algo X`` in a process of its own;
    a child that dies, times out or prints no line costs nothing: None comes back and the caller
    runs the workload in-process.  A child that prints its line is taken at its word, and the
    line says which process measured it."""
    import argparse
    import subprocess

    import bench

    args = argparse.Namespace(seed=3, allow_lib_override=False)
    calls = []

    class Done:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    def fake_run(cmd, **kw):
        calls.append(cmd)
        assert "--no-also" in cmd and "--no-cpu-baseline" in cmd and cmd[cmd.index("--seed") + 1] == "3"
        assert not any(k in kw["env"] for k in ("RANK", "WORLD_SIZE", "PFRL_BENCH_CHILD"))
        return fake_run.result

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("WORLD_SIZE", "1")
    fake_run.result = Done(1, b"")
    assert bench.also_in_own_process(args, ["--algo", "rainbow"]) is None
    fake_run.result = Done(0, b"RCCL banner\nnot json\n")
    assert bench.also_in_own_process(args, ["--algo", "rainbow"]) is None
    line = json.dumps({"metric": "m", "value": 9900.0, "config": {"workload": "w"}}).encode()
    fake_run.result = Done(0, b"noise\n" + line + b"\n")
    got = bench.also_in_own_process(args, ["--algo", "rainbow", "--steps", "50"])
    assert got["value"] == 9900.0
    assert got["config"]["process"] == "its own: bench.py --algo rainbow --steps 50"

    def raising(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, 1)

    monkeypatch.setattr(subprocess, "run", raising)
    assert bench.also_in_own_process(args, ["--algo", "sac"]) is None
    assert len(calls) == 3
