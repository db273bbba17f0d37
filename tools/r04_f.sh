#!/bin/bash
# Round-4 sixth GPU pass: gpurun --timeout 1500 -- 'bash tools/r04_f.sh r04f'
set -u
TAG=${1:-r04f}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py tests/test_bench_path_parity.py 2>&1 | tail -6 ) > $O/gpu_tests.txt
python $R/tools/layer_bench.py --batches 16384 --iters 5 --only wgrad --splits 16:1024,8:2048,32:1024,16:4096,4:4096,64:1024 > $O/layer_splits.txt 2>&1
python $R/tools/layer_bench.py --batches 16384 --iters 5 > $O/layer_16384.txt 2>&1
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40 > $O/bench_dqn.json 2> $O/bench_dqn.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- \
    python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_under_rocprof.json 2>/dev/null
head -40 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/p2/*/*_kernel_trace.csv --window-ms 250 --top 40 > $O/ppo_trace_summary.txt 2>&1
rm -rf /tmp/p2
tail -3 $O/gpu_tests.txt
cat $O/layer_splits.txt $O/layer_16384.txt
for f in bench_ppo bench_dqn; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("update_us"))
except Exception as e: print("$f", "FAILED", e)
EOF
done
head -14 $O/ppo_trace_summary.txt | cut -c1-140
