#!/bin/bash
# Round-4 Rainbow evidence, run ON THE GPU BOX: gpurun --timeout 1500 -- 'bash tools/r04_final_rainbow.sh'
set -u
R=$(pwd)
O=$R/gpurun_out/r04fin
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python -m pytest -x -q --tb=short -m gpu tests/test_hip_kernels.py tests/test_exact_sizes.py tests/test_bench_path_parity.py tests/test_agent_parity.py tests/test_replay_buffers.py -k "tree or prior or per or rainbow or sample or categorical" 2>&1 | tail -n 4 ) > $O/gpu_tests_tree.txt
tail -n 2 $O/gpu_tests_tree.txt
# 1. sampler phases: round-2 LDS sampler, lean, lean + prefetching wave (debug build, in-kernel clocks / events)
for m in lds lean prefetch; do
  ( PFRL_TREE_SAMPLE=$m timeout 300 python $R/tools/per_dbg2.py ) 2>&1 | grep "per draw" | tail -n 2
done > $O/per_sampler_phases.txt
cat $O/per_sampler_phases.txt
# 2. one-update timeline (kernel trace) and the event timeline of the pipeline
rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -- \
    python $R/bench.py --algo rainbow --no-cpu-baseline --steps 6 --warmup 2 --capacity 200000 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p3/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/rainbow_update_timeline.txt 2>&1
rm -rf /tmp/p3
tail -n 3 $O/rainbow_update_timeline.txt
( timeout 300 python $R/tools/pipeline_events.py --algo rainbow --updates 128 ) > $O/rainbow_pipeline_events.txt 2>&1
head -n 3 $O/rainbow_pipeline_events.txt | tail -n 2
# 3. HBM traffic of the 32-entry gather (separate --pmc passes)
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
        python $R/bench.py --algo rainbow --steps 4 --warmup 2 --capacity 100000 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_rainbow.json 2> $O/pmc_rainbow.err
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
head -c 600 $O/pmc_rainbow.json; echo
# 4. lines
python $R/bench.py --algo rainbow --no-cpu-baseline > $O/bench_rainbow.json 2> $O/bench_rainbow.err
python -c "
import json; d=json.loads(open('$O/bench_rainbow.json').read().strip().splitlines()[-1]); print('rainbow', d['value'], d['ms_per_step'])"
