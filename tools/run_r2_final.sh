python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r2_gpu_final.log 2>&1; tail -4 gpurun_out/r2_gpu_final.log
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; tail -3 gpurun_out/r2_bench_final.err
STEPS=100 bash tools/profile_cfg.sh c2 r2 > gpurun_out/r2_prof_c2.log 2>&1; tail -1 gpurun_out/r2_prof_c2.log
STEPS=8 bash tools/profile_cfg.sh c3 r2 > gpurun_out/r2_prof_c3.log 2>&1; tail -1 gpurun_out/r2_prof_c3.log
STEPS=6 bash tools/profile_cfg.sh c4 r2 > gpurun_out/r2_prof_c4.log 2>&1; tail -1 gpurun_out/r2_prof_c4.log
