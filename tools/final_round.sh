#!/bin/bash
# the round's closing GPU call (run through gpurun from the repo root): the two-rank tests of this build's new paths, the
# rocprofv3 profiles of the three benched configs on THIS build (tests/test_profiles_current.py), then the default bench line.
# Every part under its own timeout; what is finished is under gpurun_out/ whatever happens to the rest.
# tools/composite_ranks.patch, when present, is the newest library change as a patch: taken out again on the box when its tests
# fail (round 5: they passed — the file is gone again).
# usage: tools/final_round.sh <tag>
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
date +%s > gpurun_out/final_t0
timeout 330 python -m pytest tests/test_gpu_dist_sampler.py -x -q -m gpu -k "several_kinds or single_rank_trajectory" > gpurun_out/final_tests.log 2>&1
RC=$?
echo "tests rc=$RC t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))" | tee -a gpurun_out/final_tests.log
if [ $RC -ne 0 ] && [ -f tools/composite_ranks.patch ]; then
  # the build's newest library change (tools/composite_ranks.patch) did not pass: the profiles below are then taken on the
  # sources WITHOUT it (the caller reverts the same change in the repository, so that kernels_sha stays the tree's)
  patch -R -p1 < tools/composite_ranks.patch > gpurun_out/final_revert.log 2>&1 && python -c "from bpp_amd import build; build.build(force=True)" >> gpurun_out/final_revert.log 2>&1
  echo "reverted rc=$? sha=$(python tools/src_hash.py) t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))" | tee -a gpurun_out/final_tests.log
fi
for c in c2 c4 c3; do
  timeout 360 bash tools/profile_cfg.sh $c $TAG > gpurun_out/final_profile_$c.log 2>&1
  echo "profile $c rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
done
cd $R
timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
echo "bench rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
cp -f bench_full.json gpurun_out/bench_full.json 2>/dev/null
tail -c 600 gpurun_out/bench_default.json
