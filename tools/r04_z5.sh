#!/bin/bash
set -u
TAG=${1:-r04z5}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -q -m gpu tests 2>&1 | tail -n 12 ) > $O/gpu_tests.txt
tail -n 4 $O/gpu_tests.txt
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
( timeout 300 python $R/tools/host_delay_probe.py --algo rainbow ) > $O/host_delay_rainbow.txt 2>&1
tail -n 5 $O/host_delay_rainbow.txt
python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 60 > $O/dqn.json 2> $O/dqn.err; python -c "
import json; d=json.load(open('$O/dqn.json')); print('dqn', d['value'], d['ms_per_step'])"
