timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernel_variants" 2>&1 | tail -4
for kern in pipe pipemfma; do
  echo "== $kern"
  BPA_S20_KERNEL=$kern timeout 600 python bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline 2>gpurun_out/c4_$kern.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['roofline'])"
  grep -i "kernel\|us/launch" gpurun_out/c4_$kern.err | tail -3
done
