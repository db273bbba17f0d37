#!/bin/bash
# SAC: launch-count work (loss gradients in the forward launch, MFMA action gradient, narrow-head backward)
set -u
TAG=${1:-r04x}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_actor_kernels.py tests/test_mfma_linear.py tests/test_agent_parity.py -k "sac or head or squashed or actor or linear or twin or loss or td3 or ddpg" 2>&1 | tail -8 ) > $O/gpu_tests.txt
tail -8 $O/gpu_tests.txt
B="python $R/bench.py --algo sac --no-cpu-baseline"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-800:])
"; }
run sac_new X=1
run sac_x X=2
run sac_new2 X=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -- python $R/bench.py --algo sac --no-cpu-baseline --steps 6 --warmup 2 --capacity 200000 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p3/*/*_kernel_trace.csv --marker k_adam --every 3 > $O/sac_update_timeline.txt 2>&1
tail -50 $O/sac_update_timeline.txt
rm -rf /tmp/p3
