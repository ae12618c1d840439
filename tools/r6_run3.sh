#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_gsampler.py tests/test_gtr_posterior.py tests/test_finetune_adaptation.py tests/test_gpu_dist_sampler.py tests/test_gpu_bench_dist.py -x -q -m gpu --durations=12 > gpurun_out/r6_tests3.log 2>&1; echo "tests rc=$?"; tail -22 gpurun_out/r6_tests3.log
SH="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for v in "-" "0"; do
  if [ "$v" = "-" ]; then unset BPA_GS_FUSEA; else export BPA_GS_FUSEA=$v; fi
  for c in "c3 1250" "c3 2500"; do set -- $c
    timeout 300 python bench.py --config $1 --loci $2 $SH --full-record gpurun_out/share_$1_$2_$v.json 2> gpurun_out/share_$1_$2_$v.err | tail -1 > /dev/null
    python -c "
import json; d=json.load(open('gpurun_out/share_$1_$2_$v.json')); s=d['device_resident_sampler']; print('share $1 $2 loci [FUSEA=$v]', s['iterations_per_s'], 'it/s; launches/it', s['launches_per_iteration'])"
  done
done
unset BPA_GS_FUSEA
bash tools/timeline.sh c3 > gpurun_out/timeline_c3_r6a.txt 2>&1; cat gpurun_out/timeline_c3_r6a.txt
bash tools/trace_share.sh c3 1250 > gpurun_out/trace_share_c3_r6a.txt 2>&1; cat gpurun_out/trace_share_c3_r6a.txt
