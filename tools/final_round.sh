#!/bin/bash
# the round's closing GPU call (run through gpurun from the repo root): the rocprofv3 profiles of the three benched configs on THIS
# build (tests/test_profiles_current.py: their kernels_sha must be the tree's), the GPU time lines of a sampler iteration of configs
# 3 / 4 on the same build, then the default bench line.  Every part under its own timeout; what is finished is under gpurun_out/
# whatever happens to the rest.
# usage: tools/final_round.sh <tag>
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
date +%s > gpurun_out/final_t0
# a step that overruns its deadline is ended WITH its children (its own process group: a bench left running under rocprofv3 would
# share the GPU with everything after it — round 6's first closing call measured a 4.8x slower mixed set that way)
deadline() {   # deadline <seconds> <log> <command...>
  local secs=$1 log=$2; shift 2
  setsid "$@" > $log 2>&1 &
  local pid=$! t=0
  while kill -0 $pid 2>/dev/null; do
    sleep 2; t=$((t + 2))
    if [ $t -ge $secs ]; then kill -TERM -- -$pid 2>/dev/null; sleep 3; kill -KILL -- -$pid 2>/dev/null; wait $pid 2>/dev/null; return 124; fi
  done
  wait $pid
}
for c in c2 c4 c3; do
  PROFILE_DEADLINE_S=${PROFILE_DEADLINE_S:-420} deadline 600 gpurun_out/final_profile_$c.log bash tools/profile_cfg.sh $c $TAG
  echo "profile $c rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
done
cd $R
# the bench line below quotes `traffic` from the newest committed profiles: on this box those are the ones just taken (the caller
# copies the same files into profiles/$TAG/ of the repository)
mkdir -p profiles/$TAG
for c in c2 c3 c4; do [ -f gpurun_out/profile_${c}_$TAG.json ] && cp -f gpurun_out/profile_${c}_$TAG.json profiles/$TAG/profile_$c.json; done
for c in c3 c4; do
  deadline 300 gpurun_out/timeline_$c.txt bash tools/timeline.sh $c
  echo "timeline $c rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
done
cd $R
timeout 480 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
echo "bench rc=$? t=$(( $(date +%s) - $(cat gpurun_out/final_t0) ))"
cp -f bench_full.json gpurun_out/bench_full.json 2>/dev/null
grep "section" gpurun_out/bench_default.log | cut -c1-120
tail -c 600 gpurun_out/bench_default.json
