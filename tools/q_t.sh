timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gsampler.py tests/test_gpu_tape.py tests/test_gpu_host_driver.py tests/test_gpu_params.py "tests/test_gpu_fullsize.py::test_full_size_properties[C3-10000-1000-8-gtr-4-taus1]" -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p3 -o p -- python /root/repo/bench.py --config c3 --steps 6 --warmup 1 --no-cpu-baseline > /tmp/c3.json 2> /tmp/c3.err
python3 - <<'PY'
import csv, glob, json
f = glob.glob('/tmp/p3/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3, 2), r["Percentage"])
j = json.loads(open('/tmp/c3.json').read().strip().split('\n')[-1])
print("c3 sampler", j["value"], "tape", j["likelihood_only"]["iterations_per_s"])
PY
