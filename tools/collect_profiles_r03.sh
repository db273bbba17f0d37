#!/bin/bash
# Round-3 profile set, run ON THE GPU BOX: gpurun --timeout 1800 -- 'bash tools/collect_profiles_r03.sh'
set -u
R=$(pwd)
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. DQN headline: kernel stats + timeline + one-update timeline (final build)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- \
    python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 50 > $O/bench_under_rocprof.json 2>/dev/null
head -60 /tmp/p1/*/*_kernel_stats.csv > $O/dqn_bench_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/p1/*/*_kernel_trace.csv --window-ms 300 --top 30 > $O/dqn_bench_timeline.txt
python $R/tools/update_timeline.py /tmp/p1/*/*_kernel_trace.csv --marker k_rmsprop --every 1 > $O/dqn_update_timeline.txt
python $R/tools/trace_slice.py /tmp/p1/*/*_kernel_trace.csv --ms 16 > $O/dqn_step_slice.txt
rm -rf /tmp/p1
# 2. HBM traffic of the gathers (separate --pmc passes)
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
        python $R/bench.py --steps 4 --warmup 2 --capacity 100000 --no-cpu-baseline --no-also --no-data-path-only > /dev/null 2>&1
done
python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_gather.json 2> $O/pmc_gather.err
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
# 3. PPO: kernel stats + PMC
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- \
    python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_under_rocprof.json 2>/dev/null
head -60 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
rm -rf /tmp/p2
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
        python $R/bench.py --algo ppo --steps 128 --warmup 128 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_ppo.json 2> $O/pmc_ppo.err
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
# 4. Rainbow / SAC: one-update timelines
rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -- \
    python $R/bench.py --algo rainbow --no-cpu-baseline --steps 6 --warmup 2 --capacity 200000 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p3/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/rainbow_update_timeline.txt 2>&1
rm -rf /tmp/p3
rocprofv3 --kernel-trace --output-format csv -d /tmp/p4 -- \
    python $R/bench.py --algo sac --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p4/*/*_kernel_trace.csv --marker k_adam --every 3 > $O/sac_update_timeline.txt 2>&1
rm -rf /tmp/p4
# 5. the bench lines
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python $R/bench.py --algo rainbow --no-cpu-baseline > $O/bench_rainbow.json 2>/dev/null
python $R/bench.py --algo sac --no-cpu-baseline > $O/bench_sac.json 2>/dev/null
python $R/bench.py --host-env --no-cpu-baseline --no-also --no-data-path-only > $O/bench_hostenv.json 2>/dev/null
ls -la $O
