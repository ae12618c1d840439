cd /root/repo
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for L in 1250 2500; do
for v in - 0; do
  if [ "$v" = "-" ]; then unset BPA_GS_FUSEA; else export BPA_GS_FUSEA=$v; fi
  python bench.py --config c3 --loci $L $B --full-record /tmp/ab_env.json 2>/dev/null | tail -1 > /dev/null
  python -c "
import json; d=json.load(open('/tmp/ab_env.json')); s=d['device_resident_sampler']; print('c3 loci $L BPA_GS_FUSEA=$v', s['iterations_per_s'], 'it/s; launches/it', s['launches_per_iteration'])"
done; done
