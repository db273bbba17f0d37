#!/bin/bash
# Round-4 profile set, run ON THE GPU BOX: gpurun --timeout 2400 -- 'bash tools/collect_profiles_r04.sh'
set -u
R=$(pwd)
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. DQN headline: kernel stats + timeline + one-update timeline
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- \
    python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 50 > $O/bench_under_rocprof.json 2>/dev/null
head -60 /tmp/p1/*/*_kernel_stats.csv > $O/dqn_bench_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/p1/*/*_kernel_trace.csv --window-ms 300 --top 30 > $O/dqn_bench_timeline.txt
python $R/tools/update_timeline.py /tmp/p1/*/*_kernel_trace.csv --marker k_rmsprop_fused --every 1 > $O/dqn_update_timeline.txt
rm -rf /tmp/p1
# 2. HBM traffic of the gathers (separate --pmc passes), tagged with the gather sources' hash
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
        python $R/bench.py --steps 4 --warmup 2 --capacity 100000 --no-cpu-baseline --no-also --no-data-path-only > /dev/null 2>&1
done
python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_gather.json 2> $O/pmc_gather.err
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
# 3. PPO: kernel stats, timeline of one captured update, window summary, PMC of its gathers
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- \
    python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_under_rocprof.json 2>/dev/null
head -60 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/p2/*/*_kernel_trace.csv --window-ms 200 --top 40 > $O/ppo_trace_summary.txt 2>&1
rm -rf /tmp/p2
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- \
        python $R/bench.py --algo ppo --steps 128 --warmup 128 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_ppo.json 2> $O/pmc_ppo.err
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
# 4. per-layer table of the tile programs at update / acting / minibatch size
python $R/tools/layer_bench.py --batches 16384,512,32 --iters 10 > $O/layer_final.txt 2>&1
# 5. the data-parallel update chain on one GPU (single-rank communicator)
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40 > $O/bench_dp_single_rank.json 2> $O/bench_dp_single_rank.err
PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- \
    python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 10 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/kt/*/*_kernel_trace.csv --marker k_rmsprop_fused --every 1 > $O/dp_update_timeline.txt 2>&1
rm -rf /tmp/kt
# 6. Rainbow / SAC one-update timelines + lines
rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -- \
    python $R/bench.py --algo rainbow --no-cpu-baseline --steps 6 --warmup 2 --capacity 200000 > /dev/null 2>&1
python $R/tools/update_timeline.py /tmp/p3/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/rainbow_update_timeline.txt 2>&1
rm -rf /tmp/p3
python $R/bench.py --algo rainbow --no-cpu-baseline > $O/bench_rainbow.json 2>/dev/null
python $R/bench.py --algo sac --no-cpu-baseline > $O/bench_sac.json 2>/dev/null
python $R/bench.py --host-env --no-cpu-baseline --no-also --no-data-path-only > $O/bench_hostenv.json 2>/dev/null
# 7. the reference's PPO at full rollout size on this box's host cores
HIP_VISIBLE_DEVICES="" PFRL_REFERENCE=$R/oracle/_ref python $R/tools/reference_cpu_baseline.py --algo ppo --num-envs 512 --ppo-steps 128 \
    --threads 16 --out $O/reference_cpu_baseline_ppo_gpubox.json > /dev/null 2> $O/reference_ppo.err
# 8. the driver's line
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
ls -la $O
