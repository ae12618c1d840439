#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 480 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
echo "bench rc=$?"; cp -f bench_full.json gpurun_out/bench_full.json 2>/dev/null; tail -c 300 gpurun_out/bench_default.json; echo
SOAK_PROGRAM=1 timeout 900 python tools/soak_gsampler.py > gpurun_out/soak_program_r6.txt 2>&1; echo "soak rc=$?"; tail -12 gpurun_out/soak_program_r6.txt
