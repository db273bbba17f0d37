#!/bin/bash
# Round-4 fifth GPU pass: gpurun --timeout 1500 -- 'bash tools/r04_e.sh r04e'
set -u
TAG=${1:-r04e}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py tests/test_bench_path_parity.py \
    tests/test_agent_parity.py -k "not cartpole" 2>&1 | tail -15 ) > $O/gpu_tests.txt
python $R/tools/layer_bench.py --sweep --batches 16384 --iters 5 > $O/layer_sweep_16384.txt 2>&1
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
PFRL_PPO_UPDATE_GRAPH=0 python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_noupdgraph.json 2> $O/bench_ppo_noupdgraph.err
timeout 300 python $R/tools/ppo_time.py > $O/ppo_time.txt 2>&1
timeout 300 python $R/tools/host_profile_algo.py --algo ppo > $O/host_profile_ppo.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- \
    python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_under_rocprof.json 2>/dev/null
head -40 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/p2/*/*_kernel_trace.csv --window-ms 250 --top 40 > $O/ppo_trace_summary.txt 2>&1
rm -rf /tmp/p2
cat $O/gpu_tests.txt | tail -8
cat $O/layer_sweep_16384.txt
for f in bench_ppo bench_ppo_noupdgraph; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"])
except Exception as e: print("$f", "FAILED", e)
EOF
done
tail -4 $O/bench_ppo.err
tail -12 $O/ppo_time.txt
head -3 $O/host_profile_ppo.txt
