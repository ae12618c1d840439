python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r2_gpu5.log 2>&1; tail -6 gpurun_out/r2_gpu5.log
STEPS=100 bash tools/profile_cfg.sh c2 r2 > gpurun_out/r2_prof_c2.log 2>&1; tail -3 gpurun_out/r2_prof_c2.log
