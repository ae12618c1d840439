tools/f64_rate.bin > gpurun_out/f64_rate.txt 2>&1; cat gpurun_out/f64_rate.txt
bash tools/profile_cfg.sh c4 r3 > gpurun_out/prof_c4.log 2>&1; tail -2 gpurun_out/prof_c4.log
bash tools/profile_cfg.sh c4 r3_pipemfma BPA_S20_KERNEL=pipemfma > gpurun_out/prof_c4m.log 2>&1; tail -2 gpurun_out/prof_c4m.log
