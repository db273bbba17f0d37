#!/bin/bash
# Round-5 profile set, run ON THE GPU BOX: gpurun --timeout 2400 -- 'bash tools/collect_profiles_r05.sh [parts]'
# parts: any of  dqn pmc ppo rainbow sac line  (default: all)
set -u
PARTS=${1:-"dqn pmc ppo rainbow sac line"}
R=$(pwd)
O=$R/gpurun_out/r05
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
  python $R/tools/update_timeline.py /tmp/p1/*/*_kernel_trace.csv --marker k_rmsprop_fused --every 1 > $O/dqn_update_timeline.txt
  rm -rf /tmp/p1
fi
if has pmc; then
  # HBM traffic of the gathers (separate --pmc passes, kernel-trace only), tagged with the gather sources' hash
  pmc() { name=$1; shift
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- $B "$@" > /dev/null 2>&1
    done; }
  # PMC_ONLY="ppo" (any of gather ppo rainbow sac) restricts the passes; default: all four
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
    CAL=$(python -c "import json; print(json.load(open('$R/profiles/r05_pmc_gather.json'))['fetch_calibration_factor'])")
    python $R/tools/pmc_gather.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE --sac 14336 $CAL > $O/pmc_sac.json 2> $O/pmc_sac.err
    rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
  fi
fi
if has ppo; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- $B --algo ppo > $O/bench_ppo_under_rocprof.json 2>/dev/null
  head -60 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
  grep -h "k_gae_scan\|k_adv_" /tmp/p2/*/*_kernel_stats.csv >> $O/ppo_kernel_stats.csv
  python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker FusedAdam --every 1 > $O/ppo_update_timeline.txt 2>&1
  rm -rf /tmp/p2
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
  for n in gather ppo rainbow sac; do [ -s $O/pmc_$n.json ] && cp $O/pmc_$n.json $R/profiles/r05_pmc_$n.json; done
  python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
fi
ls -la $O
