#!/bin/bash
# full DQN step: kernel trace summary + update timeline.  gpurun -- 'bash tools/r03_full.sh r03c'
set -u
TAG=${1:-r03c}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-also \
    --no-data-path-only --steps 20 --warmup 3 --capacity 200000 > $O/bench_kt.json 2> $O/bench_kt.err
python $R/tools/update_timeline.py /tmp/kt/*/*_kernel_trace.csv --marker k_rmsprop --every 1 > $O/dqn_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/kt/*/*_kernel_trace.csv --window-ms 100 --top 30 > $O/dqn_trace_summary.txt 2>&1
python $R/tools/trace_slice.py /tmp/kt/*/*_kernel_trace.csv --ms 18 > $O/dqn_trace_slice.txt 2>&1
rm -rf /tmp/kt
head -30 $O/dqn_trace_summary.txt
