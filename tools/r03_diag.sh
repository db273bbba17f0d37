#!/bin/bash
# Round-3 diagnostics, run ON THE GPU BOX: host profile of the data-path step, step phases,
# kernel-by-kernel timeline of one DQN update, PER sampler phase clocks.
#   gpurun --timeout 1200 -- 'bash tools/r03_diag.sh r03a'
set -u
TAG=${1:-r03a}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/host_profile.py > $O/host_profile.txt 2>&1
python $R/tools/dqn_step_time.py > $O/dqn_step_time.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-also \
    --no-data-path-only --steps 10 --warmup 3 --capacity 200000 > $O/bench_kt.json 2> $O/bench_kt.err
python $R/tools/update_timeline.py /tmp/kt/*/*_kernel_trace.csv --marker k_rmsprop --every 1 > $O/dqn_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/kt/*/*_kernel_trace.csv --window-ms 80 --top 30 > $O/dqn_trace_summary.txt 2>&1
rm -rf /tmp/kt
timeout 300 python $R/tools/per_dbg.py > $O/per_dbg.txt 2>&1
tail -3 $O/per_dbg.txt
