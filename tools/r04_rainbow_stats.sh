#!/bin/bash
# rocprofv3 kernel statistics of the Rainbow bench (final build of the round)
set -u
R=$(pwd)
O=$R/gpurun_out/r04stats
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- \
    python $R/bench.py --algo rainbow --no-cpu-baseline --steps 40 --warmup 3 > $O/bench_under_rocprof.json 2>/dev/null
head -n 45 /tmp/p5/*/*_kernel_stats.csv > $O/rainbow_kernel_stats.csv
python $R/tools/update_timeline.py /tmp/p5/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/rainbow_update_timeline.txt 2>&1
rm -rf /tmp/p5
head -n 14 $O/rainbow_kernel_stats.csv | cut -c1-150
tail -n 2 $O/rainbow_update_timeline.txt
