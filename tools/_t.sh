cd /root/repo
timeout 1500 python -m pytest -x -q -m gpu tests 2>&1 | tail -12
cd /tmp
timeout 900 python /root/repo/bench.py > /root/repo/gpurun_out/r05_bench_default_a.json 2> /root/repo/gpurun_out/r05_bench_default_a.err
python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/r05_bench_default_a.json'))
print('dqn', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['mfma'].get('update_us'), d['roofline']['mfma'].get('parity_ceiling_env_steps_s'))
for k, v in d.get('also', {}).items():
    print(k, v['value'], v['ms_per_step'], (v.get('cpu_baseline') or {}).get('value'))
print('cpu', d['cpu_baseline']['value'], d['config'].get('native_lib'))
print('dpo', d.get('data_path_only', {}).get('value'), d.get('data_path_only', {}).get('without_update_launches'))
PY
