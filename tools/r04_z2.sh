#!/bin/bash
set -u
TAG=${1:-r04z2}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( PFRL_TREE_SAMPLE=paths timeout 300 python $R/tools/per_dbg2.py ) > $O/per_dbg_paths.txt 2>&1
( PFRL_TREE_SAMPLE=lds timeout 300 python $R/tools/per_dbg2.py ) > $O/per_dbg_lds.txt 2>&1
tail -5 $O/per_dbg_paths.txt $O/per_dbg_lds.txt
( cd $R && timeout 900 python -m pytest -q -m gpu tests/test_mfma_trunk.py tests/test_powf_glibc.py tests/test_reference_examples.py tests/test_reference_suite.py tests/test_replay_buffers.py tests/test_teacher_forced_loss.py 2>&1 | grep -v Warning | tail -40 ) > $O/gpu_tests_rest.txt
grep -n "Error\|passed\|failed\|assert" $O/gpu_tests_rest.txt | tail -20
