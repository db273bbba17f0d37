#!/bin/bash
# Round-4 fourth GPU pass: gpurun --timeout 1500 -- 'bash tools/r04_d.sh r04d'
set -u
TAG=${1:-r04d}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py tests/test_bench_path_parity.py \
    tests/test_teacher_forced_loss.py tests/test_fused_optimizer.py tests/test_episodic_recurrent.py \
    tests/test_agent_parity.py -k "not cartpole" 2>&1 | tail -15 ) > $O/gpu_tests.txt
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
PFRL_PPO_ACT_GRAPH=0 python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_nograph.json 2> $O/bench_ppo_nograph.err
PFRL_TRUNK_NHWC_FC_MIN_BATCH=0 python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_planar.json 2> $O/bench_ppo_planar.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- \
    python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_under_rocprof.json 2>/dev/null
head -40 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/p2/*/*_kernel_trace.csv --window-ms 280 --top 40 > $O/ppo_trace_summary.txt 2>&1
rm -rf /tmp/p2
python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40 > $O/bench_dqn.json 2> $O/bench_dqn.err
cat $O/gpu_tests.txt
for f in bench_ppo bench_ppo_nograph bench_ppo_planar bench_dqn; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("update_us"))
except Exception as e: print("$f", "FAILED", e)
EOF
done
tail -5 $O/bench_ppo.err
head -12 $O/ppo_trace_summary.txt | cut -c1-140
