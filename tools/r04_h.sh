#!/bin/bash
# Round-4 eighth GPU pass: gpurun --timeout 2400 -- 'bash tools/r04_h.sh r04h'
# whole GPU suite; the data-parallel update chain on one GPU (single-rank RCCL communicator)
set -u
TAG=${1:-r04h}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest -x -q -m gpu tests 2>&1 | tail -25 ) > $O/gpu_tests.txt
B="python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40"
$B > $O/bench_single.json 2> $O/bench_single.err
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force $B > $O/bench_dp_lowrank.json 2> $O/bench_dp_lowrank.err
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 $B > $O/bench_dp_allreduce.json 2> $O/bench_dp_allreduce.err
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force PFRL_GRAPH_COLLECTIVE=0 $B > $O/bench_dp_split.json 2> $O/bench_dp_split.err
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_SOURCES=0 PFRL_DP_LOWRANK=0 PFRL_GRAPH_COLLECTIVE=0 $B > $O/bench_dp_r03plan.json 2> $O/bench_dp_r03plan.err
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- $B > /dev/null 2> $O/dp_prof.err
python $R/tools/update_timeline.py /tmp/kt/*/*_kernel_trace.csv --marker k_rmsprop --every 1 > $O/dp_update_timeline.txt 2>&1
rm -rf /tmp/kt
tail -12 $O/gpu_tests.txt
for f in bench_single bench_dp_lowrank bench_dp_allreduce bench_dp_split bench_dp_r03plan; do python - <<EOF
import json
try:
    d=json.load(open("$O/$f.json")); print("$f",d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("update_us"))
except Exception as e:
    print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
EOF
done
tail -30 $O/dp_update_timeline.txt | cut -c1-120
