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
# the config's data set is made ONCE (bpp_amd/synth.py reads BPP_AMD_SYNTH_CACHE; c3's 10 000 GTR loci take ~20 s of numpy per
# bench start otherwise, seven starts below); PROFILE_DEADLINE_S: counter passes that would start after it are left out — the
# JSON then says which (a box's budget can end a call: better a profile without the last SQ sets than none)
export BPP_AMD_SYNTH_CACHE=/tmp/synth_cache_prof; mkdir -p $BPP_AMD_SYNTH_CACHE
T0=$(date +%s); DEADLINE=${PROFILE_DEADLINE_S:-100000}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", "$R/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from bpp_amd import synth
c = b.CONFIGS["$CFG"]
synth.precompute(c["loci"], c["sites"], c["taxa"], c["model"], c["rate_cats"], 12345, c.get("divergence", 1.0))
PY
SKIPPED=""
pass() {   # pass <name> <rocprofv3 arguments...>
  local name=$1; shift
  if [ $(( $(date +%s) - T0 )) -gt $DEADLINE ]; then SKIPPED="$SKIPPED $name"; return; fi
  rocprofv3 "$@" -f csv -d $W/$name -o p -- $BENCH > /dev/null 2> $W/$name.log
  echo "[profile_cfg] $CFG $name done at +$(( $(date +%s) - T0 ))s"
}
BENCH="python $R/bench.py --full-record $W/full.json --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection"
echo "[profile_cfg] $CFG data set ready at +$(( $(date +%s) - T0 ))s"
$BENCH --full-record $W/full_plain.json > $W/bench_plain.json 2> $W/plain.log
echo "[profile_cfg] $CFG plain run done at +$(( $(date +%s) - T0 ))s"
rocprofv3 --kernel-trace --stats -f csv -d $W/trace -o p -- $BENCH > $W/bench_trace.json 2> $W/trace.log
echo "[profile_cfg] $CFG trace pass done at +$(( $(date +%s) - T0 ))s"
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES
pass pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass pmc_sq3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64
export SKIPPED ELAPSED=$(( $(date +%s) - T0 ))
python3 - <<PY
import csv, collections, glob, json
import subprocess
import os
out = {"config": "$CFG", "tag": "$TAG", "env": "$*", "command": "$BENCH", "passes_left_out": os.environ.get("SKIPPED", "").split(), "elapsed_s": int(os.environ.get("ELAPSED", "0")),
       "kernels_sha": subprocess.run(["python3", "$R/tools/src_hash.py"], capture_output=True, text=True).stdout.strip()}
try:
    out["bench_unprofiled"] = json.loads(open("$W/bench_plain.json").read().strip().splitlines()[-1])
    out["bench_unprofiled_full"] = json.load(open("$W/full_plain.json"))
except Exception as e:
    out["bench_unprofiled"] = str(e)
ks = glob.glob("$W/trace/**/*kernel_stats.csv", recursive=True)
out["kernel_stats"] = [r for r in csv.DictReader(open(ks[0]))][:8] if ks else None
WANT = ("partials", "step_", "pmatrix", "reduce", "sweep", "decide", "iter_kernel", "gstep", "eigen", "gdec")
# Bytes and time of a kernel must share a denominator (round 5: the per-dispatch MEAN over full-batch tape launches and half-batch
# sampler launches was divided by a full launch's time).  Every figure below is over the dispatches of the kernel's LARGEST grid
# only — the full-batch launches — and the trace pass gives the mean duration of exactly those (key full_batch).
def grid_of(r):
    for k in ("Grid_Size", "Grid_Size_X"):
        if r.get(k) not in (None, ""):
            try: return int(float(r[k]))
            except ValueError: pass
    return 0
full = {}
for f in glob.glob("$W/trace/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if any(w in k for w in WANT):
            acc[k][grid_of(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))*1e-3)
    for k, d in acc.items():
        g = max(d)
        allv = [x for v in d.values() for x in v]
        full[k] = {"grid": g, "dispatches": len(d[g]), "mean_us": round(sum(d[g])/len(d[g]), 3), "all_grids_dispatches": len(allv), "all_grids_mean_us": round(sum(allv)/len(allv), 3)}
out["full_batch"] = full
pm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob("$W/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][grid_of(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, bygrid in acc.items():
            if any(w in k for w in WANT):
                g = max(bygrid)
                pm.setdefault(k, {}).update({c: {"mean": sum(v)/len(v), "dispatches": len(v), "grid": g} for c, v in bygrid[g].items()})
out["pmc_per_dispatch"] = pm
out["pmc_buckets"] = "largest grid of each kernel only (full-batch launches); full_batch[kernel].mean_us is the trace pass's mean over the same launches"
# moved bytes / time of the SAME launches, per kernel that has both counters
same = {}
for k, c in pm.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and k in full:
        b = (2*c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"])*1024
        same[k] = {"moved_bytes": round(b), "mean_us": full[k]["mean_us"], "TBps": round(b/(full[k]["mean_us"]*1e-6)/1e12, 3), "frac_of_8TBps": round(b/(full[k]["mean_us"]*1e-6)/8e12, 4)}
out["moved_same_file"] = same
json.dump(out, open("$R/gpurun_out/profile_${CFG}_$TAG.json", "w"), indent=1)
print("wrote profile_${CFG}_$TAG.json")
PY
