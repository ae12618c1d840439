#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 480 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
echo "bench rc=$?"; cp -f bench_full.json gpurun_out/bench_full.json 2>/dev/null; grep section gpurun_out/bench_default.log | tail -1; tail -c 200 gpurun_out/bench_default.json; echo
