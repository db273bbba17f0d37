#!/bin/bash
set -u
TAG=${1:-r04q}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_hip_kernels.py -k "act_head" tests/test_bench_path_parity.py -k "ppo" tests/test_agent_parity.py -k "ppo or a2c" 2>&1 | tail -6 ) > $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo2.json 2> $O/bench_ppo2.err
for f in bench_ppo bench_ppo2; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("frac"))
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
EOF
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python $R/bench.py --algo ppo --no-cpu-baseline > /dev/null 2>&1
python $R/tools/trace_summary.py /tmp/p2/*/*_kernel_trace.csv --window-ms 205 --top 45 > $O/ppo_trace_summary.txt 2>&1
python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
head -40 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
rm -rf /tmp/p2
head -16 $O/ppo_trace_summary.txt | cut -c1-150
