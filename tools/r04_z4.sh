#!/bin/bash
set -u
TAG=${1:-r04z4}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( timeout 300 python $R/tools/pipeline_events.py --algo rainbow --updates 128 ) > $O/pipeline_rainbow.txt 2>&1
tail -n 40 $O/pipeline_rainbow.txt
( GPU_MAX_HW_QUEUES=8 timeout 300 python $R/tools/pipeline_events.py --algo rainbow --updates 128 ) > $O/pipeline_rainbow_q8.txt 2>&1
head -n 3 $O/pipeline_rainbow_q8.txt | tail -n 2
B="python $R/bench.py --algo rainbow --no-cpu-baseline --steps 100 --capacity 200000"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
"; }
run rb_now X=1
run rb_q8 GPU_MAX_HW_QUEUES=8
run rb_q2 GPU_MAX_HW_QUEUES=2
