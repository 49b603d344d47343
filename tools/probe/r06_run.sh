python -m pytest tests/test_job_gpu.py -x -q -s -k "headline" 2>&1 | grep -v "^$" | tail -8
