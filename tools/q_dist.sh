cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bench_dist.py tests/test_gpu_sampler.py tests/test_gpu_dist_sampler.py -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
B="python bench.py --no-tape --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --steps 20 --warmup 3"
last() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['device_resident_sampler']['iterations_per_s'], d['ms_per_iteration'], d['device_resident_sampler']['launches_per_iteration'])"; }
for n in 1250 2500 5000 10000; do
  $B --loci $n > gpurun_out/q_$n.json 2>/dev/null; last gpurun_out/q_$n.json "one GPU, $n loci, persistent:"
  BENCH_FORCE_DIST=1 $B --loci $n > gpurun_out/q_${n}d.json 2>/dev/null; last gpurun_out/q_${n}d.json "one-rank RCCL group (native callback), $n loci, hybrid:"
done
BENCH_FORCE_DIST=1 BENCH_PY_ALLREDUCE=1 $B > gpurun_out/q_pyd.json 2>/dev/null; last gpurun_out/q_pyd.json "one-rank RCCL group (python callback), 10000 loci:"
