#!/bin/bash
set -u
TAG=${1:-r04zh}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest -x -q --tb=short -m gpu tests/test_exact_sizes.py tests/test_bench_path_parity.py tests/test_agent_parity.py tests/test_teacher_forced_loss.py -k "prior or per or rainbow or categorical or c51" 2>&1 | tail -n 12 ) > $O/gpu_tests.txt
grep -v "Warning\|warnings.warn\|^$" $O/gpu_tests.txt | tail -n 8
B="python $R/bench.py --algo rainbow --no-cpu-baseline"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
"; }
run rb_chain X=1
run rb_nochain PFRL_CHAIN_ON_REPLAY_STREAM=0
run rb_chain2 X=1
( timeout 400 python $R/tools/pipeline_events.py --algo rainbow --updates 128 --capacity 1000000 ) > $O/pipeline_rainbow_1e6.txt 2>&1
grep "updates\|tree_sample\|tree_update\|graph\|batch_exp" $O/pipeline_rainbow_1e6.txt | tail -n 12
