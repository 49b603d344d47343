#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_modules_gpu.py -q -x -p no:cacheprovider -s -k "trainkeys_all or average" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-600
