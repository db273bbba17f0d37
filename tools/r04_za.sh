#!/bin/bash
set -u
TAG=${1:-r04za}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest -x -q --tb=short -m gpu tests/test_exact_sizes.py tests/test_bench_path_parity.py tests/test_agent_parity.py tests/test_teacher_forced_loss.py tests/test_hip_kernels.py -k "rainbow or categorical or c51 or noisy or dueling or Categorical" 2>&1 | tail -n 30 ) > $O/gpu_tests.txt
grep -v "Warning\|warnings.warn\|^$" $O/gpu_tests.txt | tail -n 12
B="python $R/bench.py --algo rainbow --no-cpu-baseline --steps 100 --capacity 200000"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
"; grep -i "warn\|eager\|error" $O/$name.err | head -n 3; }
run rb_ahead X=1
run rb_noahead PFRL_NOISY_AHEAD=0
( timeout 300 python $R/tools/pipeline_events.py --algo rainbow --updates 128 ) > $O/pipeline_rainbow.txt 2>&1
grep "graph0\|updates" $O/pipeline_rainbow.txt | head -n 6
