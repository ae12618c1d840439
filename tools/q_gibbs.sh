timeout 900 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu -k "bpp_proposal_kernel or persistent_kernel_equals" 2>&1 | tail -15
timeout 800 python bench.py > gpurun_out/r3_bench_eff.json 2> gpurun_out/r3_bench_eff.err; tail -c 300 gpurun_out/r3_bench_eff.err
