#!/bin/bash
# where an iteration of the generic device sampler goes on the GPU's time line: rocprofv3 --kernel-trace of a short bench run,
# the last 40 % of the dispatches (the timed region) reduced to: wall span, union of busy time, idle gaps, per-kernel time and
# how much of it ran while another kernel ran.   usage: tools/timeline.sh <config> [ENV=VAL ...]   (through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-c3}; shift 1
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
CMD="python $R/bench.py --config $CFG --steps 6 --warmup 2 --no-tape --no-scale-projection --no-other-configs --no-cpu-baseline --no-host-control --no-bpp-program --no-efficiency --full-record /tmp/tl_full.json"
timeout 400 rocprofv3 --kernel-trace -f csv -d /tmp/tl -o p -- $CMD > /tmp/tl.out 2>&1
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in rows))
n = len(ev); ev = ev[int(0.6*n):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
# union of busy intervals
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in ev: per[k][0] += e - s; per[k][1] += 1
tot = sum(v[0] for v in per.values())
d = json.load(open("/tmp/tl_full.json"))["device_resident_sampler"]
print(f"{len(ev)} dispatches over {1e-6*(t1-t0):.2f} ms wall; busy (union) {1e-6*busy:.2f} ms = {busy/(t1-t0):.2f}; sum of kernel times {1e-6*tot:.2f} ms (overlap factor {tot/busy:.2f}); sampler {d['iterations_per_s']} it/s, {d['ms_per_iteration']} ms/iteration, {d['launches_per_iteration']} launches")
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {k:50s} {c:6d} x {1e-3*t/c:8.1f} us = {100*t/tot:5.1f} % of kernel time")
PY
