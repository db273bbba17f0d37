#!/bin/bash
# Round-4 second GPU pass: gpurun --timeout 1200 -- 'bash tools/r04_b.sh r04b'
# the large-tile programs: bit-equality test, per-layer sweep; the new round-4 GPU tests; where the
# host time of a PPO rollout step goes.
set -u
TAG=${1:-r04b}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py \
    "tests/test_fused_optimizer.py" \
    "tests/test_episodic_recurrent.py" 2>&1 | tail -15 ) > $O/gpu_tests.txt
python $R/tools/layer_bench.py --sweep --batches 16384 --iters 5 > $O/layer_sweep_16384.txt 2>&1
python $R/tools/layer_bench.py --sweep --batches 2048,512 --iters 20 > $O/layer_sweep_small.txt 2>&1
timeout 300 python $R/tools/host_profile_algo.py --algo ppo > $O/host_profile_ppo.txt 2>&1
timeout 300 python $R/tools/ppo_time.py > $O/ppo_time.txt 2>&1
python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo.json 2> $O/bench_ppo.err
cat $O/gpu_tests.txt $O/layer_sweep_16384.txt
tail -5 $O/ppo_time.txt
