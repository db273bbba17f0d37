#!/usr/bin/env python
"""Run one of the REFERENCE's example scripts, unmodified, on top of pfrl_amd (build container
only): ``import pfrl`` inside the script resolves to ``pfrl_amd`` and ``import gym`` to the
test-only shim (which knows CartPole).  A drop-in demonstration, not a benchmark.

    python tools/run_reference_example.py examples/gym/train_dqn_gym.py \\
        --env CartPole-v0 --steps 2000 --gpu -1 --outdir /tmp/out --eval-interval 1000
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import run_reference_tests as _r  # noqa: E402


def main(argv):
    script = os.path.join(_r.REFERENCE, argv[0])
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "_gymshim"))
    sys.meta_path.insert(0, _r._Redirect())
    import pfrl_amd  # noqa: F401

    for name, module in list(sys.modules.items()):
        if name == "pfrl_amd" or name.startswith("pfrl_amd."):
            sys.modules.setdefault("pfrl" + name[len("pfrl_amd"):], module)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv[1:])
