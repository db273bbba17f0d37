"""Parts of ``bench.py`` (the driver contract stays in the file at the repo root):

* ``workloads``  -- the agents / envs of BASELINE.json's configs, as the reference's examples build them
* ``roofline``   -- algorithmic bytes and FLOPs of a step, the ``roofline`` object of the line
* ``baselines``  -- the ``cpu_baseline`` legs (the reference from oracle/_ref, the C-oracle port) and
  the zero-FLOP data-path measurement
* ``supervisor`` -- N > 1: one supervised child per rank, the plan ladder, the stall watchdog; the
  ``also`` workloads in processes of their own
"""
