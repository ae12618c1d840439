#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_packing.py tests/test_gpu_parity.py tests/test_gpu_tape.py tests/test_gpu_fullsize.py tests/test_gpu_gsampler.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r6_tests13.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_tests13.log
tools/r6_ab.sh onetrip c3
timeout 200 python tools/probe_klane.py 2>&1 | tail -6
