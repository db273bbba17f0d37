#!/bin/bash
# which shapes of captured collectives survive a real peer (shared device, socket transport)?
set -u
TAG=${1:-r05pat}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
for P in A B C E F D; do
  timeout 120 python $R/tools/rccl_multirank_check.py --world 2 --shared-device --graph-pattern $P --out $O/pattern_$P.json > $O/pattern_$P.log 2>&1
  echo "pattern $P rc=$?"; grep -h "ok (\|captured\|Fatal\|Error\|assert" $O/pattern_$P.log | head -12
done
