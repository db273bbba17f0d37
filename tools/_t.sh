timeout 900 python -m pytest -x -q -m gpu tests/test_hip_kernels.py -k "noisy" 2>&1 | tail -30
timeout 900 python -m pytest -x -q -m gpu tests/test_bench_path_parity.py -k "noise_draws or rainbow" 2>&1 | tail -8
cd /tmp
for L in 0 1; do
PFRL_NOISY_IN_LOADER=$L timeout 600 python /root/repo/bench.py --algo rainbow --no-cpu-baseline --capacity 200000 --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in_loader=$L', d['value'], d['ms_per_step'])"
done
