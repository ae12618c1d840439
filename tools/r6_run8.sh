#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BPP_AMD_SYNTH_CACHE=/tmp/synth_cache_prof; mkdir -p $BPP_AMD_SYNTH_CACHE
B="python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --full-record /tmp/x.json"
run() { # name, env..., then -- extra bench flags
  local name=$1; shift
  local t0=$(date +%s)
  ( export "$@"; timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/pm_$name -o p -- $B $EXTRA > /tmp/pm_$name.out 2> /tmp/pm_$name.log ); local rc=$?
  echo "== $name rc=$rc $(( $(date +%s) - t0 ))s"; grep "\[bench" /tmp/pm_$name.log | tail -4 | cut -c1-160
}
EXTRA="--no-sampler" run tapeonly X=1
EXTRA="--no-tape" run sampler_dev X=1
EXTRA="--no-tape" run sampler_host BPA_GS_HOSTDEC=1
