#!/bin/bash
# rocprofv3 profile of bench.py for one config on the GPU box: kernel trace + separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE / SQ+MFMA), reduced to one small JSON under gpurun_out/.
# usage: tools/profile_cfg.sh <config> <tag> [ENV=VAL ...]   (run through gpurun from the repo root)
set -u
CFG=${1:-c4}; TAG=${2:-r1}; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/prof_${CFG}_$TAG
rm -rf $W; mkdir -p $W $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-6}
BENCH="python $R/bench.py --full-record $W/full.json --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection"
$BENCH --full-record $W/full_plain.json > $W/bench_plain.json 2> $W/plain.log
rocprofv3 --kernel-trace --stats -f csv -d $W/trace -o p -- $BENCH > $W/bench_trace.json 2> $W/trace.log
rocprofv3 --pmc FETCH_SIZE -f csv -d $W/pmc_fetch -o p -- $BENCH > /dev/null 2> $W/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $W/pmc_write -o p -- $BENCH > /dev/null 2> $W/pmc_write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES -f csv -d $W/pmc_sq -o p -- $BENCH > /dev/null 2> $W/pmc_sq.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $W/pmc_sq2 -o p -- $BENCH > /dev/null 2> $W/pmc_sq2.log
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 -f csv -d $W/pmc_sq3 -o p -- $BENCH > /dev/null 2> $W/pmc_sq3.log
python3 - <<PY
import csv, collections, glob, json
import subprocess
out = {"config": "$CFG", "tag": "$TAG", "env": "$*", "command": "$BENCH",
       "kernels_sha": subprocess.run(["python3", "$R/tools/src_hash.py"], capture_output=True, text=True).stdout.strip()}
try:
    out["bench_unprofiled"] = json.loads(open("$W/bench_plain.json").read().strip().splitlines()[-1])
    out["bench_unprofiled_full"] = json.load(open("$W/full_plain.json"))
except Exception as e:
    out["bench_unprofiled"] = str(e)
ks = glob.glob("$W/trace/**/*kernel_stats.csv", recursive=True)
out["kernel_stats"] = [r for r in csv.DictReader(open(ks[0]))][:8] if ks else None
pm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob("$W/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if any(w in k for w in ("partials", "step_", "pmatrix", "reduce", "sweep", "decide", "iter_kernel", "gstep", "eigen")):
                pm.setdefault(k, {}).update({c: {"mean": sum(v)/len(v), "dispatches": len(v)} for c, v in d.items()})
out["pmc_per_dispatch"] = pm
json.dump(out, open("$R/gpurun_out/profile_${CFG}_$TAG.json", "w"), indent=1)
print("wrote profile_${CFG}_$TAG.json")
PY
