#!/bin/bash
# the GPU test tier + one short line per workload:  gpurun --timeout 1500 -- 'bash tools/gpu_check.sh TAG [pytest -k expr]'
set -u
TAG=${1:-r05chk}
KEXPR=${2:-}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  ( cd $R && timeout 1200 python -m pytest -x -q -m gpu tests -k "$KEXPR" 2>&1 | tail -15 ) > $O/gpu_tests.txt
else
  ( cd $R && timeout 1200 python -m pytest -x -q -m gpu tests 2>&1 | tail -15 ) > $O/gpu_tests.txt
fi
tail -8 $O/gpu_tests.txt
for A in ${ALGOS:-ppo}; do
  timeout 600 python $R/bench.py --algo $A --no-cpu-baseline ${BENCH_ARGS:-} > $O/bench_$A.json 2> $O/bench_$A.err
  python - <<PY
import json
try:
    d = json.load(open('$O/bench_$A.json')); r = d.get('roofline') or {}
    print('$A', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'mfma', (r.get('mfma') or {}).get('frac'), 'scan', json.dumps(r.get('scan_kernels'))[:600])
except Exception as e:
    print('$A FAILED', e); print(open('$O/bench_$A.err').read()[-1500:])
PY
done
