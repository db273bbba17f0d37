#!/bin/bash
# Multi-rank FUNCTIONAL run of the data plane on a ONE-GPU box (the lease refuses CPX partitioning,
# profiles/r05_cpx_refused.txt): all ranks on device 0, every rank its own NCCL_HOSTID so that RCCL
# accepts them (socket transport on loopback).   gpurun --timeout 900 -- 'bash tools/multirank_shared_device.sh r05mr'
set -u
TAG=${1:-r05mr}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
for P in B A; do
  timeout 120 python $R/tools/rccl_multirank_check.py --world 2 --shared-device --graph-pattern $P --out $O/rccl_multirank_w2_$P.json > $O/rccl_multirank_w2_$P.log 2>&1
  echo "rccl_multirank_check world=2 pattern=$P rc=$?"; grep -h "ok (\|captured\|Fatal" $O/rccl_multirank_w2_$P.log | tr '\n' ';'; echo
done
timeout 120 python $R/tools/rccl_multirank_check.py --world 4 --shared-device --graph-pattern B --out $O/rccl_multirank_w4_B.json > $O/rccl_multirank_w4_B.log 2>&1
echo "rccl_multirank_check world=4 pattern=B rc=$?"
# the agents' data-parallel update under real peers: DQN (captured collective), then the same with the
# direct data plane refused (what a failing communicator degrades to), then PPO
export PFRL_RCCL_SHARED_DEVICE=1 PFRL_BENCH_STALL_S=200
run() { name=$1; shift
  timeout 400 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT \
      $R/bench.py --gpus 2 $ARGS > $O/$name.json 2> $O/$name.err
  echo "$name rc=$?"; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print(' ', d['value'], d['ms_per_step'], {k: d['config'].get(k) for k in ('ranks_seen','dp_plan','collective_us','rccl_init_s','dp_attempts_failed')})
except Exception as e:
    print('  no line:', e); print(open('$O/$name.err').read()[-1500:])
PY
}
ARGS="--steps 6 --warmup 3 --num-envs 64 --capacity 100000 --no-cpu-baseline --no-also --no-data-path-only"
PORT=29521 run bench_dqn_w2 X=1
# a plan that takes the workers down (SIGSEGV in hipStreamEndCapture): the supervisors move on
PORT=29531 run bench_dqn_w2_fork_crash PFRL_DP_FORK_IN_CAPTURE=1
PORT=29522 run bench_dqn_w2_refused PFRL_RCCL_SHARED_DEVICE=0
PORT=29523 run bench_dqn_w2_split PFRL_FORCE_SPLIT_GRAPH=1 PFRL_GRAPH_COLLECTIVE=0
ARGS="--algo ppo --steps 128 --warmup 128 --num-envs 64 --no-cpu-baseline"
PORT=29524 run bench_ppo_w2 X=1
ARGS="--algo sac --steps 20 --warmup 10 --num-envs 16 --capacity 20000 --no-cpu-baseline"
PORT=29525 run bench_sac_w2 X=1
ls $O
