#!/bin/bash
# Round-4 third GPU pass: gpurun --timeout 1500 -- 'bash tools/r04_c.sh r04c'
set -u
TAG=${1:-r04c}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py tests/test_teacher_forced_loss.py \
    tests/test_fused_optimizer.py tests/test_episodic_recurrent.py 2>&1 | tail -15 ) > $O/gpu_tests.txt
for V in default vgpr; do
  if [ $V = default ]; then unset PFRL_AMD_LIB; else export PFRL_AMD_LIB=$R/tools/variants/libpfrl_amd_$V.so; fi
  python $R/tools/layer_bench.py --batches 16384,32 --iters 10 > $O/layer_$V.txt 2>&1
  python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40 > $O/bench_dqn_$V.json 2> $O/bench_dqn_$V.err
  python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_$V.json 2> $O/bench_ppo_$V.err
done
unset PFRL_AMD_LIB
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- \
    python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_under_rocprof.json 2>/dev/null
head -40 /tmp/p2/*/*_kernel_stats.csv > $O/ppo_kernel_stats.csv
python $R/tools/update_timeline.py /tmp/p2/*/*_kernel_trace.csv --marker multi_tensor_apply --every 1 > $O/ppo_update_timeline.txt 2>&1
python $R/tools/trace_summary.py /tmp/p2/*/*_kernel_trace.csv --window-ms 330 --top 40 > $O/ppo_trace_summary.txt 2>&1
rm -rf /tmp/p2
cat $O/gpu_tests.txt; paste $O/layer_default.txt $O/layer_vgpr.txt | cut -c1-200
for V in default vgpr; do python - <<EOF
import json
for a in ("dqn","ppo"):
    d=json.load(open("$O/bench_%s_$V.json"%a)); print("$V",a,d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("update_us"))
EOF
done
tail -3 $O/ppo_update_timeline.txt
