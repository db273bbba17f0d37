#!/bin/bash
set -u
TAG=${1:-r04zg}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest -x -q --tb=short -m gpu tests/test_hip_kernels.py tests/test_exact_sizes.py tests/test_bench_path_parity.py -k "tree or prior or per or rainbow or sample" 2>&1 | tail -n 3 )
( PFRL_TREE_SAMPLE=prefetch timeout 300 python $R/tools/per_dbg2.py ) 2>&1 | grep "per draw" | tail -n 2
( PER_DBG_LOAD=1 PFRL_TREE_SAMPLE=prefetch timeout 300 python $R/tools/per_dbg2.py ) 2>&1 | grep "per draw" | tail -n 2
( timeout 400 python $R/tools/pipeline_events.py --algo rainbow --updates 128 --capacity 1000000 ) > $O/pipeline_rainbow_1e6.txt 2>&1
grep "updates\|tree_sample\|tree_update\|graph0\|batch_exp" $O/pipeline_rainbow_1e6.txt | tail -n 9
python $R/bench.py --algo rainbow --no-cpu-baseline > $O/bench_rainbow.json 2> $O/bench_rainbow.err
python -c "
import json; d=json.loads(open('$O/bench_rainbow.json').read().strip().splitlines()[-1]); print('rainbow', d['value'], d['ms_per_step'], d['roofline'].get('traffic'))"
