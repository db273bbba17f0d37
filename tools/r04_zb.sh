#!/bin/bash
set -u
TAG=${1:-r04zb}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( timeout 400 python $R/tools/pipeline_events.py --algo rainbow --updates 128 --capacity 1000000 ) > $O/pipeline_rainbow_1e6.txt 2>&1
tail -n 30 $O/pipeline_rainbow_1e6.txt
