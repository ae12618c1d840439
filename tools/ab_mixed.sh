#!/bin/bash
# the mixed set of the bench (9 990 config-2 loci + 10 GTR+G4 loci, one composite sampler) with the generic part's per-locus
# steps as launches (BPA_GS_CHAIN=0), as ONE chain launch per sweep (=1), and by the default rule:  tools/ab_mixed.sh [nodd]
cd "$(dirname "$0")/.."
N=${1:-10}
for c in 0 1 default; do
  if [ $c = default ]; then unset BPA_GS_CHAIN; else export BPA_GS_CHAIN=$c; fi
  python - <<P
import bench, json
from bpp_amd import synth
cfg = bench.CONFIGS["c2"]
data = synth.make_dataset(cfg["loci"], cfg["sites"], cfg["taxa"], cfg["model"], cfg["rate_cats"], seed=12345)
r = bench.run_mixed_set(data, nodd=$N, iters=400)
print("BPA_GS_CHAIN=$c", r["implementation"], r["iterations_per_s"], "it/s", r["ms_per_iteration"], "ms", "acc", r["acceptance"])
P
done
