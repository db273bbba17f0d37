#!/bin/bash
set -u
TAG=${1:-r04z9}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -q --tb=short -m gpu tests 2>&1 | tail -n 80 ) > $O/gpu_tests.txt
grep -v "Warning\|warnings.warn\|^$\|capture_end\|float(log_prob)" $O/gpu_tests.txt | tail -n 30
