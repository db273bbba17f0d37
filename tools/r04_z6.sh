#!/bin/bash
set -u
TAG=${1:-r04z6}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest -x -q -m gpu tests/test_hip_kernels.py tests/test_exact_sizes.py tests/test_replay_buffers.py tests/test_bench_path_parity.py tests/test_agent_parity.py -k "tree or prior or per or rainbow or sample or categorical" 2>&1 | tail -n 5 ) > $O/gpu_tests_tree.txt
tail -n 3 $O/gpu_tests_tree.txt
( PFRL_TREE_SAMPLE=prefetch timeout 300 python $R/tools/per_dbg2.py ) > $O/per_dbg_prefetch.txt 2>&1
tail -n 2 $O/per_dbg_prefetch.txt
( timeout 300 python $R/tools/pipeline_events.py --algo rainbow --updates 128 ) > $O/pipeline_rainbow.txt 2>&1
tail -n 32 $O/pipeline_rainbow.txt
B="python $R/bench.py --algo rainbow --no-cpu-baseline --steps 100 --capacity 200000"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
"; }
run rb_late X=1
run rb_nolate PFRL_LATE_BACKWARD=0
run rb_late_lean PFRL_TREE_SAMPLE=lean
