#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_linear_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_modules_gpu.py -q -x -p no:cacheprovider -k "config4 or train or loss or harvest" 2>&1 | tail -4
timeout 600 python tools/bench_train.py --graph --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
timeout 600 python tools/probe/op_census.py --top 25 2>&1 | grep -v amdgpu.ids | tail -27
