#!/bin/bash
# census of the fine-tune step's torch ops + the tests of the kernels touched since the last full run
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/r04o_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r04o_tests.log
timeout 600 python tools/probe/op_census.py --top 120 > gpurun_out/r04o_census.log 2>&1; echo "census exit $?"; head -60 gpurun_out/r04o_census.log
