#!/bin/bash
set -u
R=$(pwd)
O=$R/gpurun_out/r04last
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2 )
python $R/bench.py --no-also --no-cpu-baseline --no-data-path-only --steps 60 > $O/bench_dqn.json 2> $O/bench_dqn.err
python -c "
import json
d=json.load(open('$O/bench_dqn.json'))
print('dqn', d['value'], d['ms_per_step'])
pl=d['roofline']['mfma'].get('per_launch')
print(pl['source'] if pl else None)
for l in (pl or {}).get('launches', []): print(' ', l['what'], l['us'], l['frac'])
"
