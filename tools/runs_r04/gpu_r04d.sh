#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python tools/bench_gemm.py qcheck8 2>&1 | grep -v amdgpu.ids | tail -15
timeout 300 python tools/bench_gemm.py qcheck 2>&1 | grep -v amdgpu.ids | tail -2
