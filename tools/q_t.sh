timeout 900 python -m pytest tests/test_gpu_host_driver.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
A00_PROF=1 timeout 600 python bench.py --no-tape --no-other-configs --no-cpu-baseline --no-efficiency --no-sampler 2>gpurun_out/hc.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(json.dumps(j.get('host_control_in_c'))[:300])"
grep "a00\]" gpurun_out/hc.err | tail -2
timeout 600 python bench.py --no-tape --no-other-configs --no-cpu-baseline --no-efficiency --no-sampler 2>gpurun_out/hc.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(json.dumps(j.get('host_control_in_c'))[:300])"
