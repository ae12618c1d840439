mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "20_state or golden" > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log)
(BPA_S20_KERNEL=waverl timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_gsampler.py -x -q -m gpu > gpurun_out/t4.log 2>&1; tail -3 gpurun_out/t4.log)
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection"
for k in wave waverl; do
  BPA_S20_KERNEL=$k python bench.py --config c4 $B --full-record gpurun_out/full_c4_$k.json 2> gpurun_out/b_c4_$k.err | tail -1 > gpurun_out/b_c4_$k.json
done
python - <<'P'
import json
for w in ('wave','waverl'):
    try:
        d=json.load(open(f'gpurun_out/full_c4_{w}.json'))
        lo=d['likelihood_only']; r=lo['roofline']; s=d['device_resident_sampler']
        print('c4',w,'tape it/s',lo['iterations_per_s'],'kernel us',r['avg_kernel_us'],'frac',r['frac'],'codes',r.get('frac_codes'),'flops',r.get('flops_frac'),'| sampler it/s',s['iterations_per_s'])
    except Exception as e: print(w,'ERR',e)
P
