B="python bench.py --no-tape --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program"
timeout 300 python tools/dbg_sweep2.py > gpurun_out/dbg2.log 2>&1; grep -c OK gpurun_out/dbg2.log; tail -1 gpurun_out/dbg2.log
BPA_SMP_DBG=48 $B --steps 100 --warmup 10 > /dev/null 2> gpurun_out/q_prof.err; grep smp2 gpurun_out/q_prof.err | tail -2
$B --steps 2000 --warmup 100 > gpurun_out/q_new.json 2> gpurun_out/q_new.err; python -c "import json;d=json.load(open('gpurun_out/q_new.json'));print(d['value'],d['ms_per_step'])"
