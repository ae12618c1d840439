#!/bin/bash
# kernel-trace stats of the generic device sampler on a share of a config: tools/trace_share.sh c3 1250
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-c3}; LOCI=${2:-1250}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs
CMD="python $R/bench.py --config $CFG --loci $LOCI --steps 60 --warmup 5 --no-tape --no-scale-projection --no-other-configs --no-cpu-baseline --no-host-control --no-bpp-program --no-efficiency"
timeout 200 $CMD 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $LOCI loci:', d['value'], 'it/s', d['ms_per_step'], 'ms/iteration')"
timeout 280 rocprofv3 --kernel-trace --stats -f csv -d /tmp/trs -o p -- $CMD > /dev/null 2>&1
f=$(find /tmp/trs -name '*kernel_stats.csv' | head -1); head -14 "$f" | cut -c1-150
