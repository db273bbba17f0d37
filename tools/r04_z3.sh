#!/bin/bash
set -u
TAG=${1:-r04z3}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( PFRL_TREE_SAMPLE=paths timeout 300 python $R/tools/per_dbg2.py ) > $O/per_dbg_lean.txt 2>&1
tail -n 4 $O/per_dbg_lean.txt
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_hip_kernels.py tests/test_exact_sizes.py tests/test_replay_buffers.py tests/test_bench_path_parity.py -k "tree or prior or per or rainbow or sample" 2>&1 | tail -n 5 ) > $O/gpu_tests_tree.txt
tail -n 3 $O/gpu_tests_tree.txt
( timeout 300 python $R/tools/host_profile_algo.py --algo rainbow ) > $O/host_profile_rainbow.txt 2>&1
head -n 40 $O/host_profile_rainbow.txt
( timeout 300 python $R/tools/host_delay_probe.py --algo rainbow ) > $O/host_delay_rainbow.txt 2>&1
tail -n 6 $O/host_delay_rainbow.txt
