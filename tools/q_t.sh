timeout 900 python -m pytest tests/test_gpu_bpp_hip.py -q -m gpu --durations=4 2>&1 | grep -E "passed|failed|Error|call|setup" | tail -8
