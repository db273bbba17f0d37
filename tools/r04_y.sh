#!/bin/bash
# DQN chain after the load-batching changes (head TD launch, RMSprop slab folds)
set -u
TAG=${1:-r04ac}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest -x -q -m gpu tests/test_fused_optimizer.py tests/test_hip_kernels.py tests/test_bench_path_parity.py tests/test_teacher_forced_loss.py tests/test_agent_parity.py -k "dqn or rmsprop or optim or td or head or loss" 2>&1 | tail -6 ) > $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
B="python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 100"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['roofline']['mfma'].get('update_us'))
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-800:])
"; }
run dqn_1 X=1
run dqn_2 X=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -- python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 30 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p1/*/*_kernel_trace.csv --marker k_rmsprop_fused --every 1 > $O/dqn_update_timeline.txt
cat $O/dqn_update_timeline.txt
rm -rf /tmp/p1
