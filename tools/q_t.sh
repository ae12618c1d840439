for nc in 0 1; do
if [ $nc = 1 ]; then export A00_NO_COHORTS=1; echo "--- A00_NO_COHORTS=1 (one engine, one batch per step)"; else echo "--- default: two cohorts on two engines"; fi
A00_PROF=1 timeout 600 python bench.py --no-tape --no-other-configs --no-cpu-baseline --no-efficiency --no-sampler 2>gpurun_out/hc.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); h=j.get('host_control_in_c'); h.pop('note',None); print(json.dumps(h))"
grep "a00\]" gpurun_out/hc.err | tail -2
done
