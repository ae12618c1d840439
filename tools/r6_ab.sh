#!/bin/bash
# round 6: tape + sampler rates of c3 / c4 with an environment switch per run (through gpurun from the repo root)
# usage: tools/r6_ab.sh <tag> <config> [ENV=VAL ...]
TAG=$1; CFG=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export BPP_AMD_SYNTH_CACHE=/tmp/synth_cache_prof; mkdir -p $BPP_AMD_SYNTH_CACHE
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection"
timeout 400 python bench.py --config $CFG $B --full-record gpurun_out/ab_${CFG}_$TAG.json 2> gpurun_out/ab_${CFG}_$TAG.err | tail -1 > /dev/null
python - <<P
import json
try:
    d = json.load(open("gpurun_out/ab_${CFG}_$TAG.json"))
    lo = d.get("likelihood_only") or {}; r = lo.get("roofline") or {}; s = d.get("device_resident_sampler") or {}
    print("$CFG $TAG [$*] tape it/s", lo.get("iterations_per_s"), "kernel us", r.get("avg_kernel_us"), "| sampler it/s", s.get("iterations_per_s"), "launches/it", s.get("launches_per_iteration"))
except Exception as e:
    print("$CFG $TAG ERR", e)
P
