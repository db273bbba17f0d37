#!/bin/bash
# Round-4 seventh GPU pass: gpurun --timeout 1800 -- 'bash tools/r04_g.sh r04g'
set -u
TAG=${1:-r04g}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest -x -q -m gpu tests/test_mfma_trunk.py "tests/test_exact_sizes.py::test_sac_configs4_at_capacity_1e6_matches_oracle" 2>&1 | tail -6 ) > $O/gpu_tests.txt
for V in default ldr40; do
  if [ $V = default ]; then unset PFRL_AMD_LIB; else export PFRL_AMD_LIB=$R/tools/variants/libpfrl_amd_$V.so; fi
  python $R/tools/layer_bench.py --batches 16384,32 --iters 10 > $O/layer_$V.txt 2>&1
  python $R/bench.py --no-cpu-baseline --no-also --no-data-path-only --steps 40 > $O/bench_dqn_$V.json 2> $O/bench_dqn_$V.err
  python $R/bench.py --algo ppo --no-cpu-baseline > $O/bench_ppo_$V.json 2> $O/bench_ppo_$V.err
done
unset PFRL_AMD_LIB
( time python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/gpu_tests.txt
paste $O/layer_default.txt $O/layer_ldr40.txt | cut -c1-200
for V in default ldr40; do python - <<EOF
import json
for a in ("dqn","ppo"):
    try:
        d=json.load(open("$O/bench_%s_$V.json"%a)); print("$V",a,d["value"],d["ms_per_step"],d.get("roofline",{}).get("mfma",{}).get("update_us"))
    except Exception as e: print("$V", a, "FAILED", e)
EOF
done
cat $O/bench_default.time
python - <<EOF
import json
d=json.load(open("$O/bench_default.json")); print(d["value"], d["ms_per_step"], d.get("cpu_baseline",{}).get("value"))
for k,v in d.get("also",{}).items(): print(k, v["value"], v["ms_per_step"], v["config"].get("prefill_s"), (v.get("cpu_baseline") or {}).get("value"), (v.get("roofline") or {}).get("mfma",{}).get("frac"))
print(d.get("data_path_only"))
EOF
tail -5 $O/bench_default.err
