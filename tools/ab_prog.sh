# the generic sampler with the program's moves on configs 3 / 4 (bench sections restricted) + the adaptation tests
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_finetune_adaptation.py -x -q -m gpu > gpurun_out/t5.log 2>&1; tail -5 gpurun_out/t5.log)
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-control --no-bpp-program --no-efficiency --no-scale-projection --no-tape"
for c in c3 c4; do
  python bench.py --config $c $B --full-record gpurun_out/full_${c}_prog.json 2> gpurun_out/b_${c}_prog.err | tail -1 > gpurun_out/b_${c}_prog.json
  tail -3 gpurun_out/b_${c}_prog.err
done
python - <<'P'
import json
for c in ('c3','c4'):
    try:
        d=json.load(open(f'gpurun_out/full_{c}_prog.json')); s=d['device_resident_sampler']
        print(c,'sampler it/s',s['iterations_per_s'],'launches/it',s['launches_per_iteration'],'acc',s['acceptance'],s.get('moves_short'),s.get('step_lengths_after_burnin'),s.get('theta_gibbs_draws_generic'))
    except Exception as e: print(c,'ERR',e)
P
