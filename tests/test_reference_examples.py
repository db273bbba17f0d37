"""The reference's own Atari example script, UNMODIFIED, on the device path (north star: "drops
into examples/atari unchanged"; reference examples/atari/train_dqn_batch_ale.py:161-283:
make_atari -> wrap_deepmind -> MultiprocessVectorEnv -> VectorFrameStack ->
train_agent_batch_with_evaluation).  ``import pfrl`` resolves to pfrl_amd; gym / ALE / OpenCV are
not installed, so tests/_gymshim supplies an ALE-shaped scripted game and the two OpenCV calls
(test side only).  The script is the reference's file where /root/reference exists, else its
compiled form from oracle/_ref (what travels to the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_script():
    rel = "examples/atari/train_dqn_batch_ale.py"
    return (os.path.exists(os.path.join("/root/reference", rel))
            or os.path.exists(os.path.join(ROOT, "oracle", "_ref", rel + "c")))


def _run_example(tmp_path, gpu, steps, num_envs):
    out = str(tmp_path / "out")
    report = str(tmp_path / "report.json")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_reference_example.py"),
           "examples/atari/train_dqn_batch_ale.py", "--pfrl-amd-report", report,
           "--gpu", str(gpu), "--num-envs", str(num_envs), "--steps", str(steps),
           "--env", "PongNoFrameskip-v4", "--arch", "nature", "--agent", "DQN",
           "--replay-start-size", "400", "--eval-interval", str(steps // 2), "--eval-n-runs", "2",
           "--target-update-interval", "400", "--max-frames", "4000", "--outdir", out,
           "--log-level", "30"]
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    runs = [d for d in os.listdir(out)]
    assert len(runs) == 1
    run = os.path.join(out, runs[0])
    return run, json.load(open(report))


@pytest.mark.gpu
@pytest.mark.skipif(not _have_script(), reason="oracle/_ref not built (build container step)")
def test_reference_atari_example_runs_unmodified_on_the_device_path(tmp_path):
    steps, n = 2000, 8
    run, rep = _run_example(tmp_path, 0, steps, n)
    # the driver's products (pfrl/experiments/evaluator.py:388-393, train_agent_batch.py:143-154)
    rows = open(os.path.join(run, "scores.txt")).read().strip().splitlines()
    assert rows[0].split("\t")[:4] == ["steps", "episodes", "elapsed", "mean"]
    assert len(rows) >= 3                                   # header + two evaluations
    assert "average_q" in rows[0] and "n_updates" in rows[0]
    assert os.path.isdir(os.path.join(run, "%d_finish" % steps))
    # ... made by the device path: replay in HBM, host LazyFrames ingested batch-wise, updates
    # replayed from captured graphs
    assert rep["agent"] == "DQN" and rep["device"] == "cuda:0" and rep["t"] == steps
    assert rep["replay_is_device"] and rep["frame_ring_device"] == "cuda:0"
    assert rep["frame_ring_bytes"] > 7 * 10 ** 9            # ReplayBuffer(10**6) of the script
    assert rep["replay_len"] == steps                       # one-step transitions, none dropped
    assert rep["ingest_many_calls"] >= steps // n - 2
    # one 84x84 frame per env step crossed PCIe (+ resets), not the 4-frame stack
    assert steps <= rep["frames_written"] <= steps + steps // 4
    assert rep["optim_t"] >= (steps - 400) // 4 - 2
    assert rep["use_graphs"] and rep["graphs_captured"] >= 1


@pytest.mark.skipif(not _have_script(), reason="oracle/_ref not built (build container step)")
def test_reference_atari_example_runs_unmodified_on_the_host_path(tmp_path):
    """gpu=-1: the plumbing path of BASELINE configs[0], same script."""
    run, rep = _run_example(tmp_path, -1, 600, 2)
    assert os.path.isdir(os.path.join(run, "600_finish"))
    assert rep["device"] == "cpu" and not rep["replay_is_device"] and rep["t"] == 600
