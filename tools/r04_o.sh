#!/bin/bash
set -u
TAG=${1:-r04o}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py tests/test_hip_kernels.py -k "large_tile or act_head or conv_kernels or nature_trunk" 2>&1 | tail -8 ) > $O/gpu_tests.txt
tail -8 $O/gpu_tests.txt
python $R/tools/layer_bench.py --sweep --batches 16384,2048 --iters 5 --only dgrad --layers conv3,conv2 > $O/layer_dgrad.txt 2>&1
cat $O/layer_dgrad.txt
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
PFRL_QNET_DGRAD=0 python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_nopos.json 2> $O/bench_ppo_nopos.err
for f in bench_ppo bench_ppo_nopos; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("frac"))
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
EOF
done
