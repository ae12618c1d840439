#!/bin/bash
# the round's closing GPU call (run through gpurun from the repo root): the two-rank tests of this build's new paths, the
# rocprofv3 profiles of the three benched configs on THIS build (tests/test_profiles_current.py), then the default bench line.
# Every part under its own timeout; what is finished is under gpurun_out/ whatever happens to the rest.
# usage: tools/final_round.sh <tag>
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
date +%s > gpurun_out/final_t0
timeout 420 python -m pytest tests/test_gpu_dist_sampler.py -x -q -m gpu -k "several_kinds or single_rank_trajectory" > gpurun_out/final_tests.log 2>&1
echo "tests rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))" | tee -a gpurun_out/final_tests.log
for c in c2 c4 c3; do
  timeout 360 bash tools/profile_cfg.sh $c $TAG > gpurun_out/final_profile_$c.log 2>&1
  echo "profile $c rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
done
cd $R
timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
echo "bench rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
cp -f bench_full.json gpurun_out/bench_full.json 2>/dev/null
tail -c 600 gpurun_out/bench_default.json
