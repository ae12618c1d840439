( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 2>&1 | grep -v "^$" | tail -45 ) > gpurun_out/r3_gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r3_gpu_tests.log | tail -2
