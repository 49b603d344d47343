#!/bin/bash
# f4 (importance sampling) tests on the GPU
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "sample_pdf or importance or rowdot or nerf_module or pose_block" > gpurun_out/r04p_tests.log 2>&1; echo "tests exit $?"; tail -40 gpurun_out/r04p_tests.log
