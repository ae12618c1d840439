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
}
BENCH="python $R/bench.py --full-record $W/full.json --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection"
$BENCH --full-record $W/full_plain.json > $W/bench_plain.json 2> $W/plain.log
rocprofv3 --kernel-trace --stats -f csv -d $W/trace -o p -- $BENCH > $W/bench_trace.json 2> $W/trace.log
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
