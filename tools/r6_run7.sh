#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist_sampler.py tests/test_gpu_bench_dist.py tests/test_gpu_gsampler.py -x -q -m gpu -k "program or eight or config3" > gpurun_out/r6_tests7.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6_tests7.log
PROFILE_DEADLINE_S=700 timeout 1000 bash tools/profile_cfg.sh c4 r6x 2>&1 | grep -E "profile_cfg|wrote|Error|error" | head -20
