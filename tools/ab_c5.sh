# config 5 (anopheles) with the program's moves: the per-locus steps as one chain launch (default) or a launch per step
(timeout 600 python -m pytest tests/test_gpu_gsampler.py tests/test_anopheles.py -x -q -m gpu 2>&1 | tail -2)
for v in chain nochain; do
  if [ $v = nochain ]; then export BPA_GS_CHAIN=0; else unset BPA_GS_CHAIN; fi
  python - <<'P'
import sys, os; sys.path.insert(0, os.getcwd())
import bench, bpp_amd
e = bpp_amd.Engine(0); r = bench.run_config5(e, iters=200); e.close()
print(os.environ.get("BPA_GS_CHAIN", "default"), r["iterations_per_s"], "it/s", r["launches_per_iteration"], "launches/it", r["acceptance"])
P
done
