#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python tools/probe/retarget_check.py 2>&1 | tail -9
timeout 900 python -m pytest tests/test_capture_gpu.py -x -q > gpurun_out/r04w_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r04w_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 --poses 2 --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
timeout 600 python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
