#!/bin/bash
# the device sampler of a config with an environment switch off and on:  tools/ab_env.sh <config> <VAR> <value> [<value> ...]
# (e.g. tools/ab_env.sh c3 BPA_GS_FUSEA 0 1; "-" = unset)
cd "$(dirname "$0")/.."
CFG=$1; VAR=$2; shift 2
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  python bench.py --config $CFG $B --full-record /tmp/ab_env.json 2>/dev/null | tail -1 > /dev/null
  python -c "
import json; d=json.load(open('/tmp/ab_env.json')); s=d['device_resident_sampler']; print('$CFG $VAR=$v', s['iterations_per_s'], 'it/s; launches/it', s['launches_per_iteration'], s.get('moves','')[:40])"
done
