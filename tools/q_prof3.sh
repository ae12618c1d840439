B="python bench.py --no-tape --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program"
BPA_SMP_DBG=112 $B --steps 100 --warmup 10 > /dev/null 2> gpurun_out/q_prof.err; grep smp2 gpurun_out/q_prof.err | tail -2
