#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BPP_AMD_SYNTH_CACHE=/tmp/synth_cache_prof; mkdir -p $BPP_AMD_SYNTH_CACHE
B="python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --full-record /tmp/x.json"
t0=$(date +%s)
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/pm_dev -o p -- $B > /tmp/pm_dev.out 2> /tmp/pm_dev.log; echo "== pmc pass, device decisions: rc=$? $(( $(date +%s) - t0 ))s"; grep "\[bench" /tmp/pm_dev.log | tail -2 | cut -c1-160
cd $R
tools/r6_ab.sh paced c3
tools/r6_ab.sh paced c4
