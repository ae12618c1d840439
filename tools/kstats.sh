#!/bin/bash
# per-kernel times of one bench.py command: tools/kstats.sh <tag> [ENV=VAL ...] -- <bench args>
TAG=$1; shift
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/ks_$TAG; rm -rf $W; mkdir -p $W $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $W -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-bpp-program --no-other-configs --no-host-control > $W/bench.json 2> $W/log
python3 - <<PY
import csv, glob
ks = glob.glob("$W/**/*kernel_stats.csv", recursive=True)
print("== $TAG")
for r in list(csv.DictReader(open(ks[0])))[:10]:
    print("%-70s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
