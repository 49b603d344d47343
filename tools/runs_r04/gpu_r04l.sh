#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_backward_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 600 python tools/bench_kernels.py gemm_tn 2>&1 | grep -v amdgpu.ids | tail -12
timeout 600 python tools/bench_train.py --graph --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_r04a.so timeout 600 python tools/bench_train.py --graph --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
