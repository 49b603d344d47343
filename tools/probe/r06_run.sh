python -m pytest tests/test_backward_gpu.py tests/test_linear_gpu.py -q -s 2>&1 | grep -a "worst gradient\|passed\|failed" | cut -c1-400
