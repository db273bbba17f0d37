#!/bin/bash
set -u
TAG=${1:-r04u}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest -x -q -m gpu tests/test_bench_path_parity.py tests/test_mfma_trunk.py tests/test_agent_parity.py -k "dqn or nhwc or trunk" tests/test_exact_sizes.py -k "dqn or configs1" 2>&1 | tail -6 ) > $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
B="python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 100"
run() { name=$1; shift; ( env "$@" $B > $O/$name.json 2> $O/$name.err ); python -c "
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['roofline']['mfma'].get('update_us'))
except Exception as e: print('$name FAILED', e); print(open('$O/$name.err').read()[-800:])
"; }
run off_1 PFRL_DQN_RANGE_OVERLAP=0
run cut25_1 PFRL_DQN_RANGE_CUT=0.25
run cut125_1 PFRL_DQN_RANGE_CUT=0.125
run cut50_1 PFRL_DQN_RANGE_CUT=0.5
run off_2 PFRL_DQN_RANGE_OVERLAP=0
run cut25_2 PFRL_DQN_RANGE_CUT=0.25
run cut25_3sets PFRL_DQN_RANGE_CUT=0.25 PFRL_MANY_SETS=3
