( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r3_gpu_tests.log 2>&1
timeout 800 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
tail -3 gpurun_out/r3_gpu_tests.log
