for flags in "--no-cpu-baseline --no-other-configs --no-efficiency" "--no-other-configs --no-efficiency --no-tape" "--no-cpu-baseline --no-efficiency --no-tape"; do
  timeout 600 python bench.py $flags 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$flags ->', (j.get('host_control_in_c') or {}).get('iterations_per_s'))"
done
