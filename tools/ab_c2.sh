# config 2's headline sampler alone (the persistent kernel with the program's moves): tests first, then three rate measurements
(timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_dist_sampler.py -x -q -m gpu 2>&1 | tail -2)
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape --no-uniform-kernel"
for i in 1 2 3; do python bench.py $B --full-record /tmp/c2_$i.json 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 it/s', d['value'], 'ms/it', d['ms_per_iteration'])"; done
