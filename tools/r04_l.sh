#!/bin/bash
set -u
TAG=${1:-r04l}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_fused_optimizer.py tests/test_distributed.py 2>&1 | tail -8 ) > $O/gpu_tests.txt
tail -8 $O/gpu_tests.txt
B="python -X faulthandler $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40"
run() { name=$1; shift; ( env PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err ); rc=$?; echo "== $name rc=$rc $(python -c "
import json,sys
try:
    d=json.load(open('$O/$name.json')); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('mfma',{}).get('update_us'))
except Exception as e: print('no json')
")"; if [ $rc != 0 ]; then grep -v "^frame" $O/$name.err | grep -A25 "Fatal Python error\|Traceback\|Error" | head -60 | cut -c1-180; fi; }
for i in 1 2 3; do run lowrank_$i PFRL_DP_LOWRANK=force; done
run early_1 PFRL_DP_LOWRANK=0
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 10 > /dev/null 2> $O/dp_prof.err
python $R/tools/update_timeline.py /tmp/kt/*/*_kernel_trace.csv --marker k_rmsprop --every 1 > $O/dp_update_timeline.txt 2>&1
rm -rf /tmp/kt
tail -25 $O/dp_update_timeline.txt | cut -c1-130
