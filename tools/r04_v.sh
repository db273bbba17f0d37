#!/bin/bash
# data path phases + kernel trace of the replay-side-only step
set -u
TAG=${1:-r04v}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python tools/data_path_phases.py > $O/phases.txt 2> $O/phases.err ); tail -4 $O/phases.txt; tail -3 $O/phases.err
timeout 900 rocprofv3 --kernel-trace -d $O/trace -o dp --output-format csv -- python $R/tools/data_path_phases.py > $O/phases_rocprof.txt 2>&1
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_slice.py $CSV --ms 1.5 > $O/slice.txt; tail -80 $O/slice.txt
python $R/tools/trace_summary.py $CSV --window-ms 20 > $O/summary.txt; head -40 $O/summary.txt
rm -rf $O/trace
