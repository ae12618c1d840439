A00_PROF=1 timeout 600 python bench.py --no-tape --no-other-configs --no-cpu-baseline --no-efficiency 2>gpurun_out/hc.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(json.dumps(j.get('host_control_in_c'))[:900])"
grep "a00\]" gpurun_out/hc.err | tail -3
