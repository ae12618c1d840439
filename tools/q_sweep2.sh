B="python bench.py --no-tape --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program"
$B --steps 2000 --warmup 100 > gpurun_out/q_new.json 2> gpurun_out/q_new.err; tail -2 gpurun_out/q_new.err; python -c "import json;d=json.load(open('gpurun_out/q_new.json'));print(d['value'],d['ms_per_step'],d['roofline'])"
BPA_SMP_DBG=16 $B --steps 100 --warmup 10 > /dev/null 2> gpurun_out/q_prof.err; grep smp2 gpurun_out/q_prof.err | tail -3
BPA_SMP_V1=1 $B --steps 500 --warmup 50 > gpurun_out/q_old.json 2> gpurun_out/q_old.err; python -c "import json;d=json.load(open('gpurun_out/q_old.json'));print(d['value'],d['ms_per_step'])"
python -m pytest tests/test_gpu_sampler.py tests/test_a00_posterior.py -x -q -p no:cacheprovider 2>&1 | tail -3
