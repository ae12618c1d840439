#!/bin/bash
# rocprofv3 profile of bench.py for one config (c3 / c4) on the GPU box; outputs under gpurun_out/
# usage: tools/profile_cfg.sh <config> <tag> [env assignments...]   (run through gpurun from the repo root)
set -u
CFG=${1:-c4}; TAG=${2:-r1}; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${CFG}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline"
$BENCH > $OUT/bench_plain.json 2> $OUT/plain.log
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o $CFG -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o $CFG -- $BENCH > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o $CFG -- $BENCH > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -f csv -d $OUT/pmc_sq -o $CFG -- $BENCH > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log
find $OUT -name "*.csv" | head -30
