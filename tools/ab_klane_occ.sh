# step_s4_klane_v3_kernel at several register budgets / chunk sizes, built ON the GPU box (hipcc is there): c3 tape kernel time
B="--config c3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-sampler"
for v in "4 4" "5 4" "5 2" "6 2" "4 2"; do set -- $v
  BPA_HIPCC_FLAGS="-DBPA_KLANE_OCC=$1 -DBPA_KLANE_CH=$2" python -m bpp_amd.build --force > /tmp/build.log 2>&1 || { echo "occ $1 ch $2: build failed"; tail -3 /tmp/build.log; continue; }
  python tools/kres.py klane_v3 | head -1 | sed 's/.*vgpr/vgpr/'
  python bench.py $B --full-record /tmp/occ.json 2>/dev/null | tail -1 > /dev/null
  python -c "
import json; d=json.load(open('/tmp/occ.json')); r=d['likelihood_only']['roofline']; print('occ $1 ch $2: kernel us', r['avg_kernel_us'], 'tape it/s', d['likelihood_only']['iterations_per_s'])"
done
