#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r6_tests4.log 2>&1; echo "tests rc=$?"; tail -16 gpurun_out/r6_tests4.log
tools/r6_ab.sh defer c3
timeout 200 python tools/probe_klane.py 2>&1 | tail -6
BPA_HIPCC_FLAGS="-DBPA_KLANE_DEFER=0" python -m bpp_amd.build --force > /tmp/build.log 2>&1 || { echo build failed; tail -3 /tmp/build.log; }
tools/r6_ab.sh nodefer c3
timeout 200 python tools/probe_klane.py 2>&1 | tail -6
