B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for v in split nosplit; do
  if [ $v = nosplit ]; then export BPA_GS_NOSPLIT=1; else unset BPA_GS_NOSPLIT; fi
  python bench.py --config c3 $B --full-record /tmp/c3_$v.json 2>/dev/null | tail -1 > /dev/null
  python -c "
import json; d=json.load(open('/tmp/c3_$v.json')); s=d['device_resident_sampler']; print('$v', s['iterations_per_s'], 'launches/it', s['launches_per_iteration'])"
done
