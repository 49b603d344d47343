#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_capture_gpu.py -x -q > gpurun_out/r04w_tests.log 2>&1; echo "tests exit $?"; tail -12 gpurun_out/r04w_tests.log | cut -c1-300
