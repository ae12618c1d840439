#!/bin/bash
# the GPU's time line of one iteration of config 5 (anopheles, generic sampler, the program's moves): rocprofv3 --kernel-trace, the dispatches of
# the last full iteration in start order with their gaps.   usage: tools/timeline_c5.sh - [ENV=VAL ...]   (through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-10}; shift 1
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl5
cat > /tmp/tl5_run.py <<P
import sys; sys.path.insert(0, "$R")
import bench, bpp_amd
e = bpp_amd.Engine(0); r = bench.run_config5(e, iters=60); e.close()
print(r["implementation"] if "implementation" in r else "", r["iterations_per_s"], "it/s")
P
timeout 400 rocprofv3 --kernel-trace -f csv -d /tmp/tl5 -o p -- python /tmp/tl5_run.py > /tmp/tl5.out 2>&1
tail -2 /tmp/tl5.out
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tl5/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
# an iteration = from one persistent sweep launch (iter_kernel) to the next
idx = [i for i, e in enumerate(ev) if "gchain_kernel" in e[2]]
if len(idx) < 4: idx = [0, len(ev) - 1]
a, b = idx[-3], idx[-2]
t0 = ev[a][0]
print(f"one iteration: {b - a} dispatches, {1e-3*(ev[b][0]-t0):.1f} us")
prev_end = t0
for s, e, k, q in ev[a:b]:
    print(f"  +{1e-3*(s-t0):8.1f} us  {1e-3*(e-s):7.1f} us  gap {1e-3*(s-prev_end):6.1f}  q{q}  {k}")
    prev_end = max(prev_end, e)
per = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in ev[a:b]: per[k][0] += e - s; per[k][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {k:60s} {c:4d} x {1e-3*t/c:7.1f} us = {1e-3*t:8.1f} us")
PY
