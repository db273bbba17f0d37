#!/bin/bash
# Round-4 closing evidence, run ON THE GPU BOX: gpurun --timeout 1500 -- 'bash tools/r04_final.sh'
set -u
R=$(pwd)
O=$R/gpurun_out/r04final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -q --tb=short -m gpu tests 2>&1 | tail -n 40 ) > $O/gpu_suite_tail.txt
grep -v "Warning\|warnings.warn\|^$\|capture_end\|float(log_prob)" $O/gpu_suite_tail.txt | tail -n 6
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json
d=json.load(open('$O/bench_default.json'))
print('dqn', d['value'], d['ms_per_step'], 'update_us', d['roofline']['mfma'].get('update_us'))
for k, v in (d.get('also') or {}).items():
    print(' also', k, v.get('value'), v.get('ms_per_step'))
print(' data_path_only', d.get('extra', {}).get('data_path_only'))
print(' cpu_baseline', d.get('cpu_baseline'))
"
python $R/bench.py --algo sac --no-cpu-baseline > $O/bench_sac.json 2>/dev/null
python $R/bench.py --host-env --no-cpu-baseline --no-also --no-data-path-only > $O/bench_hostenv.json 2>/dev/null
python -c "
import json
for n in ('bench_sac', 'bench_hostenv'):
    d=json.loads(open('$O/%s.json' % n).read().strip().splitlines()[-1]); print(n, d['value'], d['ms_per_step'])
"
