timeout 900 python -m pytest tests/test_gpu_prior.py -x -q -m gpu --durations=4 -k several 2>&1 | grep -E "passed|failed|Error|assert|call" | tail -8
