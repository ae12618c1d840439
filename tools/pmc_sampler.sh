#!/bin/bash
# SQ counters of the device-resident sampler's kernels under rocprofv3 --pmc (own passes, no tracing domains)
# usage: tools/pmc_sampler.sh <tag> [ENV=VAL ...]
set -u
TAG=${1:-x}; shift 1
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmc_smp_$TAG
mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_sampler_msc.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -f csv -d $OUT/a -o x -- $CMD > $OUT/a.out 2> $OUT/a.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $OUT/b -o x -- $CMD > $OUT/b.out 2> $OUT/b.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM -f csv -d $OUT/c -o x -- $CMD > $OUT/c.out 2> $OUT/c.log
python3 - <<PY > $R/gpurun_out/pmc_smp_$TAG.txt
import csv, collections, glob
for sub in ("a", "b", "c"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "smp" in k or "iter_kernel" in k:
                print(sub, k, {c: (round(sum(v)/len(v)), len(v)) for c, v in d.items()})
PY
tail -3 $OUT/c.log >> $R/gpurun_out/pmc_smp_$TAG.txt
cat $R/gpurun_out/pmc_smp_$TAG.txt
