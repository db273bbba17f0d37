#!/bin/bash
# DP debug matrix on one GPU (single-rank RCCL): which combination trips the NCCL watchdog
set -u
TAG=${1:-r04i}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 10 --capacity 100000"
run() { name=$1; shift; ( env PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err ); echo "$name rc=$? $(head -c 150 $O/$name.json | cut -c1-120)"; grep -m1 "watchdog thread terminated" $O/$name.err | cut -c1-200; }
run a_src1_lr0_gc0 PFRL_DP_SOURCES=1 PFRL_DP_LOWRANK=0 PFRL_GRAPH_COLLECTIVE=0
run b_src0_lrF_gc0 PFRL_DP_SOURCES=0 PFRL_DP_LOWRANK=force PFRL_GRAPH_COLLECTIVE=0
run c_src0_lr0_gc1 PFRL_DP_SOURCES=0 PFRL_DP_LOWRANK=0 PFRL_GRAPH_COLLECTIVE=1
run d_src1_lr0_gc1 PFRL_DP_SOURCES=1 PFRL_DP_LOWRANK=0 PFRL_GRAPH_COLLECTIVE=1
run e_src1_lrF_gc1 PFRL_DP_SOURCES=1 PFRL_DP_LOWRANK=force PFRL_GRAPH_COLLECTIVE=1
run f_src1_lrF_gc1_nocache PFRL_DP_SOURCES=1 PFRL_DP_LOWRANK=force PFRL_GRAPH_COLLECTIVE=1 TORCH_NCCL_CUDA_EVENT_CACHE=0
run g_auto PFRL_DP_LOWRANK=force
run h_auto_nocache PFRL_DP_LOWRANK=force TORCH_NCCL_CUDA_EVENT_CACHE=0
( cd $R && timeout 600 python -m pytest -x -q -m gpu tests/test_fused_optimizer.py -k data_parallel tests/test_distributed.py 2>&1 | tail -15 ) > $O/gpu_tests.txt
tail -15 $O/gpu_tests.txt
