#!/bin/bash
# the generic sampler's chain launch (BPA_GS_CHAIN) on / off: config 3 shares, config 5, the mixed set
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for loci in 1250 2500 5000; do for ch in 0 1; do
  BPA_GS_CHAIN=$ch timeout 200 python bench.py --config c3 --loci $loci --steps 40 --warmup 5 --no-tape --no-scale-projection --no-other-configs --no-cpu-baseline --no-host-control --no-bpp-program --no-efficiency 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 $loci loci chain=$ch:', d['value'], 'it/s')"
done; done
for ch in 0 1; do BPA_GS_CHAIN=$ch timeout 100 python -c "
import bench, bpp_amd
e = bpp_amd.Engine(0); r = bench.run_config5(e, iters=200); print('c5 chain=$ch', r['iterations_per_s'])" 2>/dev/null | tail -1; done
for ch in 0 1; do BPA_GS_CHAIN=$ch timeout 200 python tools/mixed_rate.py 2>/dev/null | sed "s/^/chain=$ch /" | tail -2; done
