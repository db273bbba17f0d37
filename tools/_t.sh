timeout 900 python -m pytest -x -q -m gpu tests/test_bench_path_parity.py -k "noise_draws or rainbow" 2>&1 | tail -30
cd /tmp
for F in 0 1; do
PFRL_NOISE_FEED=$F timeout 600 python /root/repo/bench.py --algo rainbow --no-cpu-baseline --capacity 200000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('feed=$F', d['value'], d['ms_per_step'])"
done
