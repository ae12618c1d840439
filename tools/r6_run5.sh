#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/r6_tests5.log 2>&1; echo "tests rc=$?"; tail -16 gpurun_out/r6_tests5.log
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r6a.json 2> gpurun_out/bench_r6a.log; echo "bench rc=$?"; grep "section\|dataset" gpurun_out/bench_r6a.log | cut -c1-150; tail -c 400 gpurun_out/bench_r6a.json; echo
cp -f bench_full.json gpurun_out/bench_full_r6a.json 2>/dev/null
BPA_EXPERIMENTAL=1 python -m bpp_amd.build --force > /tmp/build_exp.log 2>&1 || { echo "experimental build failed"; tail -5 /tmp/build_exp.log; }
python -c "import bpp_amd; print('experimental build:', bpp_amd.lib().bpa_experimental_build())"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gsampler.py -x -q -m gpu -k "variants or parameter_moves or program_s_moves" > gpurun_out/r6_tests5_exp.log 2>&1; echo "experimental tests rc=$?"; tail -5 gpurun_out/r6_tests5_exp.log
