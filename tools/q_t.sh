timeout 900 python -m pytest tests/test_gpu_host_driver.py tests/test_gpu_edges.py tests/test_gpu_packing.py tests/test_gpu_gsampler.py tests/test_gpu_bigsampler.py tests/test_gpu_tape.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do
timeout 600 python bench.py --no-tape --no-other-configs --no-cpu-baseline --no-efficiency --no-sampler 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); h=j.get('host_control_in_c'); h.pop('note',None); print(json.dumps(h))"
done
