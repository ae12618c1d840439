# A/B of the round-5 step kernels against round 4's on the GPU box: parity tests first, then tape + sampler rates per config
# usage: bash tools/ab_kernels.sh [configs...]   (through gpurun from the repo root)
mkdir -p gpurun_out
CFGS=${@:-c3 c4}
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tape.py tests/test_gpu_edges.py tests/test_gpu_gsampler.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log)
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection"
for c in $CFGS; do
  python bench.py --config $c $B --full-record gpurun_out/full_${c}_new.json 2> gpurun_out/b_${c}_new.err | tail -1 > gpurun_out/b_${c}_new.json
  BPA_KLANE_V2=1 BPA_S20_KERNEL=pipe python bench.py --config $c $B --full-record gpurun_out/full_${c}_old.json 2> gpurun_out/b_${c}_old.err | tail -1 > gpurun_out/b_${c}_old.json
done
python - $CFGS <<'P'
import json, sys
for c in sys.argv[1:]:
    for w in ('old','new'):
        try:
            d=json.load(open(f'gpurun_out/full_{c}_{w}.json'))
            lo=d['likelihood_only']; r=lo['roofline']; s=d['device_resident_sampler']
            print(c,w,'tape it/s',lo['iterations_per_s'],'kernel us',r['avg_kernel_us'],'frac',r['frac'],'codes',r.get('frac_codes'),'flops',r.get('flops_frac'),'| sampler it/s',s['iterations_per_s'])
        except Exception as e: print(c,w,'ERR',e)
P
