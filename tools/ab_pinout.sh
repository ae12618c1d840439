#!/bin/bash
# the sums of the program's moves on the generic sampler written straight into pinned host memory (default) or into a device
# buffer and copied (BPA_GS_PINOUT=0): config 5 (anopheles) and config 3
cd "$(dirname "$0")/.."
for v in 0 1 -; do
  if [ "$v" = "-" ]; then unset BPA_GS_PINOUT; else export BPA_GS_PINOUT=$v; fi
  python - <<'P'
import sys, os; sys.path.insert(0, os.getcwd())
import bench, bpp_amd
e = bpp_amd.Engine(0); r = bench.run_config5(e, iters=300); e.close()
print("c5 BPA_GS_PINOUT=" + os.environ.get("BPA_GS_PINOUT", "default"), r["iterations_per_s"], "it/s", r["launches_per_iteration"], "launches/it", r["acceptance"])
P
done
tools/ab_env.sh c3 BPA_GS_PINOUT 0 1 - 2>&1 | tail -3
