#!/bin/bash
set -u
TAG=${1:-r04r}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_hip_kernels.py -k "act_head" tests/test_bench_path_parity.py -k "ppo" 2>&1 | tail -3 )
python $R/tools/head_time.py 2>&1 | tail -4
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo2.json 2> $O/bench_ppo2.err
for f in bench_ppo bench_ppo2; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("frac"))
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
EOF
done
