#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6_tests1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_tests1.log
tools/r6_ab.sh skip c3
tools/r6_ab.sh store c3 BPA_GS_ROOTSTORE=1
tools/r6_ab.sh skip c4
tools/r6_ab.sh store c4 BPA_GS_ROOTSTORE=1
timeout 300 python tools/probe_klane.py > gpurun_out/probe_klane_r6a.txt 2>&1; tail -12 gpurun_out/probe_klane_r6a.txt
