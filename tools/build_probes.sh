#!/bin/bash
# the micro-benchmarks DESIGN.md quotes (launch floor, MFMA rates / layout / accumulation order): gfx950 executables,
# cross-compiled here, run on the GPU box through gpurun (e.g. gpurun -- tools/mfma_probe.bin)
set -e
cd "$(dirname "$0")"
for f in probe2 probe_bw mfma_order f64_rate; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o $f.bin $f.hip
done
ls -la *.bin
