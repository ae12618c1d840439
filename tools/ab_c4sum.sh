B="--steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for v in fused launch; do
  if [ $v = launch ]; then export BPA_S20_SUM_LAUNCH=1; else unset BPA_S20_SUM_LAUNCH; fi
  python bench.py --config c4 $B --full-record gpurun_out/full_c4_$v.json 2> gpurun_out/b_c4_$v.err | tail -1 > /dev/null
  python -c "
import json; d=json.load(open('gpurun_out/full_c4_$v.json')); s=d['device_resident_sampler']; print('$v', s['iterations_per_s'], 'launches/it', s['launches_per_iteration'])"
done
