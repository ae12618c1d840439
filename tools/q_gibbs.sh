timeout 900 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu -k "bpp_proposal_kernel or persistent_kernel_equals" 2>&1 | tail -5
timeout 600 python tools/q_prog.py 2>&1 | grep -v "^\[bpp_amd\] smp2" | tail -8
BPA_SMP_DBG=16 timeout 600 python tools/q_prog.py program 2>&1 | tail -30
