#!/usr/bin/env python
"""Run one of the REFERENCE's example scripts, unmodified, on top of pfrl_amd: ``import pfrl``
inside the script resolves to ``pfrl_amd``, ``import gym`` / ``import cv2`` to the test-only
stand-ins under tests/_gymshim (CartPole, an ALE-shaped scripted game).  The script itself comes
from /root/reference where that exists and otherwise from oracle/_ref (its compiled form, built
by oracle/build_ref.py; what travels to the GPU box).  A drop-in demonstration, not a benchmark.

    python tools/run_reference_example.py examples/gym/train_dqn_gym.py \\
        --env CartPole-v0 --steps 2000 --gpu -1 --outdir /tmp/out --eval-interval 1000
    python tools/run_reference_example.py examples/atari/train_dqn_batch_ale.py --gpu 0 \\
        --num-envs 8 --steps 2000 --env PongNoFrameskip-v4 ... [--pfrl-amd-report report.json]
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import run_reference_tests as _r  # noqa: E402


def _script_path(rel):
    src = os.path.join(_r.REFERENCE, rel)
    if os.path.exists(src):
        return src
    compiled = os.path.join(ROOT, "oracle", "_ref", rel + "c")
    if os.path.exists(compiled):
        return compiled
    raise SystemExit("neither %s nor %s exists (run oracle/build_ref.py in the build container)"
                     % (src, compiled))


def main(argv):
    report = None
    if "--pfrl-amd-report" in argv:
        i = argv.index("--pfrl-amd-report")
        report = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    script = _script_path(argv[0])
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "_gymshim"))
    sys.meta_path.insert(0, _r._Redirect())
    import pfrl_amd  # noqa: F401

    for name, module in list(sys.modules.items()):
        if name == "pfrl_amd" or name.startswith("pfrl_amd."):
            sys.modules.setdefault("pfrl" + name[len("pfrl_amd"):], module)
    seen = {"agents": [], "ingest_many": 0}
    if report:
        # what the run used, observed from outside the script: the agents it built and the
        # batched host-observation ingest of the device replay store
        from pfrl_amd.agents import dqn as _dqn
        from pfrl_amd.replay_buffers import device_replay as _dr

        init = _dqn.DQN.__init__

        def spy_init(self, *a, **k):
            init(self, *a, **k)
            seen["agents"].append(self)

        _dqn.DQN.__init__ = spy_init
        ingest = _dr.DeviceReplayStore.ingest_many

        def spy_ingest(self, obs_list):
            out = ingest(self, obs_list)
            seen["ingest_many"] += out is not None
            return out

        _dr.DeviceReplayStore.ingest_many = spy_ingest
    sys.argv = [script] + argv[1:]
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        if report and seen["agents"]:
            import json

            ag = seen["agents"][-1]
            st = getattr(ag.replay_buffer, "store", None)
            fr = getattr(st, "frames", None)
            with open(report, "w") as f:
                json.dump({
                    "agent": type(ag).__name__, "device": str(ag.device), "t": int(ag.t),
                    "optim_t": int(ag.optim_t), "replay_len": len(ag.replay_buffer),
                    "replay_is_device": bool(getattr(ag.replay_buffer, "is_device", False)),
                    "frame_ring_device": None if fr is None else str(fr.frames.device),
                    "frame_ring_bytes": None if fr is None else int(fr.frames.numel()),
                    "frames_written": None if fr is None else int(fr.next_seq),
                    "ingest_many_calls": int(seen["ingest_many"]),
                    "use_graphs": bool(ag.use_graphs),
                    "graphs_captured": 0 if ag._graphed is None else len(ag._graphed.graphs),
                }, f)


if __name__ == "__main__":
    main(sys.argv[1:])
