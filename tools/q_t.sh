cd /tmp && export TMPDIR=/tmp
for f in 0.5 0.4 0.33 0.25; do
export BPA_GS_SPLIT_AT=$f
python /root/repo/bench.py --config c3 --steps 8 --warmup 1 --no-cpu-baseline --no-tape > /tmp/c3.json 2> /tmp/c3.err
python3 - <<'PY'
import json, os
j = json.loads(open('/tmp/c3.json').read().strip().split('\n')[-1])
print("split at", os.environ["BPA_GS_SPLIT_AT"], "c3 sampler", j["value"])
PY
done
