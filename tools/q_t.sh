timeout 900 python -m pytest tests/test_gpu_gsampler.py tests/test_gpu_dist_sampler.py tests/test_gtr_posterior.py tests/test_gpu_bench_dist.py tests/test_gpu_prior.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
cd /tmp && export TMPDIR=/tmp
python /root/repo/bench.py --config c3 --steps 8 --warmup 1 --no-cpu-baseline > /tmp/c3.json 2> /tmp/c3.err
python3 - <<'PY'
import json
j = json.loads(open('/tmp/c3.json').read().strip().split('\n')[-1])
print("c3 sampler", j["value"], "tape", j["likelihood_only"]["iterations_per_s"], j["roofline"]["avg_kernel_us"], j["roofline"]["frac"], j.get("launches_per_iteration"))
print({k: v for k, v in j.items() if k not in ("roofline", "config")})
PY
