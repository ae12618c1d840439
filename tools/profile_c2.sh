#!/bin/bash
# rocprofv3 profile of the default bench (config C2) on the GPU box; outputs under gpurun_out/
# usage: tools/profile_c2.sh <tag>        (run through gpurun from the repo root)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o c2 -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
# counters in their own passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: one pass each)
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o c2 -- $BENCH > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o c2 -- $BENCH > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
