#!/bin/bash
# config 4 A/B: CLV plane stride padded to 128-byte lines (default) vs the first layout, plain vs streamed CLV accesses
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -k 20_state -m gpu -x -q > gpurun_out/c4_tests.txt 2>&1
tail -3 gpurun_out/c4_tests.txt
for v in "pad:pipe" "pad:pipent" "pad:pipe2" "pad:pipe2nt"; do
  lay=${v%%:*}; k=${v##*:}
  if [ $lay = nopad ]; then export BPA_NO_PLANE_PAD=1; else unset BPA_NO_PLANE_PAD; fi
  BPA_S20_KERNEL=$k python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline --no-bpp-program > gpurun_out/c4_${lay}_${k}.json 2> gpurun_out/c4_${lay}_${k}.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/c4_${lay}_${k}.json").read().strip().split("\n")[-1])
    r=d.get("roofline") or d["likelihood_only"]["roofline"]
    print("$lay $k", d["value"], r["kernel"], r["avg_kernel_us"], r["frac"])
except Exception as ex:
    print("$lay $k failed", ex)
PY
done
