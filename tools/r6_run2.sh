#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gsampler.py tests/test_gtr_posterior.py tests/test_finetune_adaptation.py tests/test_gpu_dist_sampler.py tests/test_gpu_bench_dist.py -x -q -m gpu > gpurun_out/r6_tests2.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r6_tests2.log
B="--no-tape"
for c in c3 c4; do
  tools/r6_ab.sh dev $c
  tools/r6_ab.sh host $c BPA_GS_HOSTDEC=1
done
SH="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for v in dev host; do
  [ $v = host ] && export BPA_GS_HOSTDEC=1
  for c in "c3 1250" "c4 250"; do set -- $c
    timeout 300 python bench.py --config $1 --loci $2 $SH --full-record gpurun_out/share_$1_$v.json 2> gpurun_out/share_$1_$v.err | tail -1 > /dev/null
    python -c "
import json; d=json.load(open('gpurun_out/share_$1_$v.json')); s=d['device_resident_sampler']; print('share $1 $2 loci [$v]', s['iterations_per_s'], 'it/s; launches/it', s['launches_per_iteration'])"
  done
done
