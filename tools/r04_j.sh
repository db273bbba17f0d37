#!/bin/bash
set -u
TAG=${1:-r04j}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python -X faulthandler $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 10 --capacity 100000"
run() { name=$1; shift; ( env PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err ); rc=$?; echo "== $name rc=$rc $(head -c 150 $O/$name.json | cut -c60-130)"; if [ $rc != 0 ]; then grep -A40 "Fatal Python error\|Current thread\|Thread 0x" $O/$name.err | grep -v "^frame" | head -70 | cut -c1-160; fi; }
for i in 1 2 3; do
run r03_$i PFRL_DP_SOURCES=0 PFRL_DP_LOWRANK=0 PFRL_GRAPH_COLLECTIVE=0
run a_$i PFRL_DP_SOURCES=1 PFRL_DP_LOWRANK=0 PFRL_GRAPH_COLLECTIVE=0
run e_$i PFRL_DP_SOURCES=1 PFRL_DP_LOWRANK=force PFRL_GRAPH_COLLECTIVE=1
run g_$i PFRL_DP_LOWRANK=force
done
