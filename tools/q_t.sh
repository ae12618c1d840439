timeout 900 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu -k "several_sequences or edge_sizes" 2>&1 | tail -12
