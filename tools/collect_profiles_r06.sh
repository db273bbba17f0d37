#!/bin/bash
# Round-6 profile set, run ON THE GPU BOX: gpurun --timeout 2400 -- 'bash tools/collect_profiles_r06.sh [parts]'
# parts: any of  dqn sq phase pmc ppo rank rainbow sac line  (default: all)
set -u
PARTS=${1:-"dqn sq phase pmc ppo rank rainbow sac line"}
R=$(pwd)
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
B="python $R/bench.py --no-cpu-baseline"
if has dqn; then
  # DQN headline: kernel stats + window summary + one-update timeline
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- \
      $B --no-also --no-data-path-only --steps 50 > $O/bench_under_rocprof.json 2>/dev/null
  head -60 /tmp/p1/*/*_kernel_stats.csv > $O/dqn_bench_kernel_stats.csv
  python $R/tools/trace_summary.py /tmp/p1/*/*_kernel_trace.csv --window-ms 300 --top 30 > $O/dqn_bench_timeline.txt
  python $R/tools/update_timeline.py /tmp/p1/*/*_kernel_trace.csv --marker k_rmsprop_fused --every 1 --skip 8 > $O/dqn_update_timeline.txt
  python $R/tools/trace_slice.py /tmp/p1/*/*_kernel_trace.csv --ms 7 --back-ms 20 > $O/dqn_step_slice.txt
  rm -rf /tmp/p1
fi
if has sq; then
  # SQ counters of the ten launches of the B = 32 update (two passes: the SQ block has 8 counters)
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
      --kernel-trace --output-format csv -d /tmp/sq1 -- $B --no-also --no-data-path-only --steps 6 --warmup 3 --capacity 100000 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d /tmp/sq2 -- $B --no-also --no-data-path-only --steps 6 --warmup 3 --capacity 100000 > /dev/null 2>&1
  { echo "# rocprofv3 --pmc passes over bench.py --steps 6 --capacity 100000 (per-kernel averages, tools/pmc_kernels.py); pass 1 then pass 2";
    python $R/tools/pmc_kernels.py /tmp/sq1; python $R/tools/pmc_kernels.py /tmp/sq2; } > $O/dqn_update_sq_counters.txt 2>&1
  rm -rf /tmp/sq1 /tmp/sq2
fi
if has phase; then
  # in-kernel phase clocks of the forward / fused backward tile programs (debug build of the same sources)
  ( cd $R && bash tools/build_dbg.sh > /dev/null 2>&1 )
  { echo "# tools/qnet_phase.py (debug build, wall_clock64 stamps per workgroup and phase)"; python $R/tools/qnet_phase.py; } > $O/dqn_fwd_phase.txt 2>&1
  { echo "# tools/bwd_phase.py (debug build)"; python $R/tools/bwd_phase.py; } > $O/dqn_bwd_phase.txt 2>&1
  rm -f $R/tools/libpfrl_amd_dbg.so
fi
if has pmc; then
  pmc() { name=$1; shift
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- $B "$@" > /dev/null 2>&1
    done; }
  want() { [ -z "${PMC_ONLY:-}" ] || [[ " $PMC_ONLY " == *" $1 "* ]]; }
  if want gather; then
    pmc gather --steps 4 --warmup 2 --capacity 100000 --no-also --no-data-path-only
    python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_gather.json 2> $O/pmc_gather.err
    rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
  fi
  if want ppo; then
    pmc ppo --algo ppo --steps 128 --warmup 128
    python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_ppo.json 2> $O/pmc_ppo.err
    rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
  fi
  if want rainbow; then
    pmc rainbow --algo rainbow --steps 6 --warmup 3 --capacity 100000
    python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > $O/pmc_rainbow.json 2> $O/pmc_rainbow.err
    rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
  fi
  if want sac; then
    pmc sac --algo sac --steps 20 --warmup 10 --capacity 100000
    CAL=$(python -c "import json; print(json.load(open('$O/pmc_gather.json'))['fetch_calibration_factor'])" 2>/dev/null || python -c "import json; print(json.load(open('$R/profiles/r05_pmc_gather.json'))['fetch_calibration_factor'])")
    python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE --sac 14336 $CAL > $O/pmc_sac.json 2> $O/pmc_sac.err
    rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
  fi
fi
if has ppo; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- $B --algo ppo > $O/bench_ppo_under_rocprof.json 2>/dev/null
  head -60 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
  grep -h "k_gae_scan\|k_adv_\|k_ppo_loss\|k_splitk_group\|k_grad_" /tmp/p2/*/*_kernel_stats.csv >> $O/ppo_kernel_stats.csv
  python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
  rm -rf /tmp/p2
fi
if has rank; then
  # what ONE rank of the 8-GPU job runs (bench.py also.*_rank_shape_g8), under the profiler
  export PFRL_BENCH_SOFT_EXIT=1 PFRL_DIST_ALWAYS=1 PFRL_FORCE_SPLIT_GRAPH=1 PFRL_DP_LOWRANK=force MASTER_ADDR=127.0.0.1
  MASTER_PORT=29741 rocprofv3 --kernel-trace --output-format csv -d /tmp/p5 -- $B --algo ppo --num-envs 64 --steps 128 --warmup 128 --scaling weak --no-also > $O/ppo_rank_under_rocprof.json 2>/dev/null
  python $R/tools/update_timeline.py /tmp/p5/*/*_kernel_trace.csv --marker FusedOptimizerTensorListMetadata --every 1 > $O/ppo_rank_update_timeline.txt 2>&1
  python $R/tools/trace_slice.py /tmp/p5/*/*_kernel_trace.csv --ms 0.6 --back-ms 40 --no-collapse > $O/ppo_rank_act_slice.txt 2>&1
  rm -rf /tmp/p5
  MASTER_PORT=29742 rocprofv3 --kernel-trace --output-format csv -d /tmp/p6 -- $B --algo dqn --num-envs 32 --steps 160 --warmup 40 --scaling weak --no-also --no-data-path-only > $O/dqn_rank_under_rocprof.json 2>/dev/null
  python $R/tools/trace_slice.py /tmp/p6/*/*_kernel_trace.csv --ms 1.3 --back-ms 12 --no-collapse > $O/dqn_rank_step_slice.txt 2>&1
  rm -rf /tmp/p6
  unset PFRL_BENCH_SOFT_EXIT PFRL_DIST_ALWAYS PFRL_FORCE_SPLIT_GRAPH PFRL_DP_LOWRANK
fi
if has rainbow; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- \
      $B --algo rainbow --steps 6 --warmup 2 --capacity 200000 > /dev/null 2>&1
  python $R/tools/update_timeline.py /tmp/p3/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/rainbow_update_timeline.txt 2>&1
  head -50 /tmp/p3/*/*_kernel_stats.csv > $O/rainbow_kernel_stats.csv
  rm -rf /tmp/p3
  $B --algo rainbow > $O/bench_rainbow.json 2>/dev/null
fi
if has sac; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -- \
      $B --algo sac --steps 20 --warmup 10 --capacity 100000 > /dev/null 2>&1
  python $R/tools/update_timeline.py /tmp/p4/*/*_kernel_trace.csv --marker k_adam --every 1 > $O/sac_update_timeline.txt 2>&1
  head -40 /tmp/p4/*/*_kernel_stats.csv > $O/sac_kernel_stats.csv
  rm -rf /tmp/p4
  $B --algo sac > $O/bench_sac.json 2>/dev/null
fi
if has line; then
  # the driver's line (after the PMC files of this run have been put where bench.py looks for them)
  for n in gather ppo rainbow sac; do [ -s $O/pmc_$n.json ] && cp $O/pmc_$n.json $R/profiles/r06_pmc_$n.json; done
  python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
fi
ls -la $O
