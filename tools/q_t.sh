cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
python /root/repo/bench.py --config c3 --steps 8 --warmup 1 --no-cpu-baseline --no-tape > /tmp/c3.json 2> /tmp/c3.err
python3 - <<'PY'
import json
j = json.loads(open('/tmp/c3.json').read().strip().split('\n')[-1])
print("c3 sampler", j["value"], j["roofline"]["avg_kernel_us"])
PY
done
