python -m pytest tests/test_f_rows_gpu.py tests/test_capture_gpu.py tests/test_job_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -8
