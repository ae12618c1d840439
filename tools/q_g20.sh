timeout 900 python bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline 2>gpurun_out/c4s.err | tail -1 > gpurun_out/c4s.json; tail -5 gpurun_out/c4s.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/c4s.json').read())
print(j['metric'][:80], j['value'], j['ms_per_step'])
print(json.dumps(j['roofline'])[:600])
d=j.get('device_resident_sampler') or {}
print({k:d.get(k) for k in ('iterations_per_s','ms_per_iteration','launches_per_iteration','acceptance','implementation')})
print('tape', (j.get('likelihood_only') or {}).get('iterations_per_s'))
PY
