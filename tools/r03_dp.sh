#!/bin/bash
# data-path-only step: host profile + kernel timeline.  gpurun -- 'bash tools/r03_dp.sh r03b'
set -u
TAG=${1:-r03b}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/host_profile.py > $O/host_profile.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/host_profile.py > /dev/null 2> $O/kt.err
python $R/tools/trace_summary.py /tmp/kt/*/*_kernel_trace.csv --window-ms 30 --top 30 > $O/dp_trace_summary.txt 2>&1
rm -rf /tmp/kt
head -42 $O/host_profile.txt
