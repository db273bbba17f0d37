#!/bin/bash
# Round-4 first GPU pass, run ON THE GPU BOX: gpurun --timeout 1500 -- 'bash tools/r04_a.sh r04a'
# GPU tests, the per-layer table of the tile programs at the PPO / DQN batch sizes, SQ counters of the
# same launches at B = 16384, the default bench line.
set -u
TAG=${1:-r04a}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/gpu_tests.txt
python $R/tools/layer_bench.py --batches 16384,512,32 > $O/layer_bench.txt 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/layer_bench.py --batches 16384 --iters 2 > $O/pmc1.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc1 > $O/pmc_sq.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/layer_bench.py --batches 16384 --iters 2 > $O/pmc2.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc2 > $O/pmc_sq2.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc3 -- python $R/tools/layer_bench.py --batches 16384 --iters 2 > $O/pmc3.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc3 > $O/pmc_fetch.txt 2>&1
rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
ls -la $O
cat $O/gpu_tests.txt $O/layer_bench.txt
