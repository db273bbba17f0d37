#!/bin/bash
set -u
TAG=${1:-r04n}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_hip_kernels.py -k "act_head or ppo or gae or adv" tests/test_bench_path_parity.py -k "ppo" tests/test_agent_parity.py -k "ppo or a2c" 2>&1 | tail -8 ) > $O/gpu_tests.txt
tail -8 $O/gpu_tests.txt
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
PFRL_PPO_ACT_HEAD=0 python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_nohead.json 2> $O/bench_ppo_nohead.err
timeout 300 python $R/tools/host_profile_algo.py --algo ppo > $O/host_profile_ppo.txt 2>&1
for f in bench_ppo bench_ppo_nohead; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("frac"))
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
EOF
done
head -3 $O/host_profile_ppo.txt | tail -2
