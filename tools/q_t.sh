timeout 900 python -m pytest tests/test_gpu_prior.py tests/test_gpu_tape.py -x -q -m gpu --durations=6 2>&1 | grep -E "passed|failed|Error|assert|call" | tail -9
