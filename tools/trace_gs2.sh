#!/bin/bash
# kernel-trace stats of the generic sampler on one config, group-per-locus proposal kernel vs the one-lane kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-c3}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export BPA_GS_V1=1; else unset BPA_GS_V1; fi
  rm -rf /tmp/tr$v
  timeout 280 rocprofv3 --kernel-trace --stats -f csv -d /tmp/tr$v -o p -- python $R/bench.py --config $CFG --no-tape --no-scale-projection --no-other-configs --no-cpu-baseline --no-host-control --no-bpp-program --no-efficiency > /dev/null 2>&1
  echo "== $CFG v1=$v"; f=$(find /tmp/tr$v -name '*kernel_stats.csv' | head -1); head -9 "$f" | cut -c1-200
done
