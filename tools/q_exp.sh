B="python bench.py --no-tape --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program"
run() { # tag loci
  BPA_SMP_DBG=48 BPA_SMP_NOMIX=1 $B --loci $2 --steps 200 --warmup 10 > gpurun_out/x_$1_$2_nomix.json 2> gpurun_out/x.err; grep "sweep cycles" gpurun_out/x.err | tail -1; python -c "import json;d=json.load(open('gpurun_out/x_$1_$2_nomix.json'));print('$1 loci $2 nomix', d['device_resident_sampler']['iterations_per_s'], d['ms_per_step'])"
  BPA_SMP_DBG=48 $B --loci $2 --steps 200 --warmup 10 > gpurun_out/x_$1_$2.json 2> gpurun_out/x.err; grep "cycles of lane" gpurun_out/x.err | tail -1; python -c "import json;d=json.load(open('gpurun_out/x_$1_$2.json'));print('$1 loci $2 full', d['device_resident_sampler']['iterations_per_s'], d['ms_per_step'])"
}
run w8 5000
run w8 10000
cp tools/libw4.bin bpp_amd/libbpp_amd.so
run w4 5000
run w4 10000
