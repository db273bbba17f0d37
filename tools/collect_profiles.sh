#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel statistics and HBM-traffic counters of the
# headline benchmark, summarised into gpurun_out/$1 (copy what you want judged to profiles/).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02'
set -u
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 50 > $O/bench_under_rocprof.json 2>/dev/null
cp /tmp/prof_stats/*/*_kernel_stats.csv $O/dqn_bench_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/prof_stats/*/*_kernel_trace.csv --window-ms 300 --top 30 > $O/dqn_bench_timeline.txt
rm -rf /tmp/prof_stats
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
        python $R/bench.py --steps 4 --warmup 2 --capacity 100000 --no-cpu-baseline --no-also --no-data-path-only > /dev/null 2>&1
done
python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_gather.json
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
head -c 600 $O/pmc_gather.json
