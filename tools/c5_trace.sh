#!/bin/bash
# config 5 (anopheles, 100 loci x 12 tips) on the generic device sampler: wall time per iteration next to the sum of its kernels' durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/c5.py <<PY
import sys, json; sys.path.insert(0, "$R")
import bench, bpp_amd
e = bpp_amd.Engine(0); r = bench.run_config5(e, iters=200); print(json.dumps({k: r[k] for k in ("iterations_per_s", "ms_per_iteration", "launches_per_iteration", "implementation")})); e.close()
PY
timeout 120 python /tmp/c5.py 2>&1 | tail -1
rm -rf /tmp/trc5; timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/trc5 -o p -- python /tmp/c5.py > /dev/null 2>&1
f=$(find /tmp/trc5 -name '*kernel_stats.csv' | head -1); head -12 "$f" | cut -c1-160
