#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r6_tests_final.log 2>&1; echo "tests rc=$?"; tail -10 gpurun_out/r6_tests_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
