#!/bin/bash
# config 4 A/B runs (20-state kernels): tests first, then bench.py --config c4 per variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/c4_tests.txt 2>&1
tail -3 gpurun_out/c4_tests.txt
for v in "${@:-default}"; do
  unset BPA_PMAT_WG1 BPA_S20_KERNEL BPA_NO_PLANE_PAD
  for kv in ${v//,/ }; do [ $kv = default ] || export $kv; done
  python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline --no-bpp-program > gpurun_out/c4_ab.json 2> gpurun_out/c4_ab.err
  python - "$v" <<'PY'
import json, sys
try:
    d=json.loads(open("gpurun_out/c4_ab.json").read().strip().split("\n")[-1])
    r=d.get("roofline") or d["likelihood_only"]["roofline"]
    print(sys.argv[1], d["value"], r["kernel"], r["avg_kernel_us"], r["frac"])
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
done
