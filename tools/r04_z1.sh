#!/bin/bash
# Rainbow: forked no-grad passes, path-parallel sampler, update_errors fused with the pending writes
set -u
TAG=${1:-r04z1}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests 2>&1 | tail -15 ) > $O/gpu_tests.txt
tail -15 $O/gpu_tests.txt
B="python $R/bench.py --algo rainbow --no-cpu-baseline --steps 100 --capacity 200000"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
"; }
run rb_base PFRL_C51_FORK=0 PFRL_TREE_SAMPLE=lds PFRL_TREE_FUSE_ERRORS=0
run rb_fork PFRL_C51_FORK=1 PFRL_TREE_SAMPLE=lds PFRL_TREE_FUSE_ERRORS=0
run rb_all PFRL_C51_FORK=1 PFRL_TREE_SAMPLE=paths PFRL_TREE_FUSE_ERRORS=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -- \
    python $R/bench.py --algo rainbow --no-cpu-baseline --steps 6 --warmup 2 --capacity 200000 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p3/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/rainbow_update_timeline.txt 2>&1
rm -rf /tmp/p3
cat $O/rainbow_update_timeline.txt | tail -90
