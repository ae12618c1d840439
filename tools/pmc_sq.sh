#!/bin/bash
# SQ wait/active breakdown of one bench config under rocprofv3 --pmc (own pass, no tracing domains)
# usage: tools/pmc_sq.sh <config> <tag> [ENV=VAL ...]
set -u
CFG=${1:-c4}; TAG=${2:-x}; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${CFG}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES -f csv -d $OUT/a -o x -- $BENCH > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM -f csv -d $OUT/b -o x -- $BENCH > $OUT/b.json 2> $OUT/b.log
python3 - <<PY
import csv, collections, glob
for sub in ("a", "b"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "partials" in k or "step_" in k:
                print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
