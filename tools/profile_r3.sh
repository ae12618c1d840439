bash tools/profile_cfg.sh c2 r3 > gpurun_out/prof_c2.log 2>&1; tail -1 gpurun_out/prof_c2.log
bash tools/profile_cfg.sh c3 r3 > gpurun_out/prof_c3.log 2>&1; tail -1 gpurun_out/prof_c3.log
bash tools/profile_cfg.sh c4 r3 > gpurun_out/prof_c4.log 2>&1; tail -1 gpurun_out/prof_c4.log
BPA_SMP_DBG=48 timeout 300 python tools/sampler_modes.py uniform 2>&1 | grep "smp2\|uniform" | cut -c1-1500 > gpurun_out/smp2_phases.txt
BPA_SMP_DBG=16 timeout 300 python tools/sampler_modes.py program 2>&1 | grep "smp2\|program" | cut -c1-1500 >> gpurun_out/smp2_phases.txt
