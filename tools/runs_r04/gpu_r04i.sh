#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
echo "== default"; timeout 600 python tools/bench_gemm.py time 2>&1 | grep -v amdgpu.ids | cut -c1-160
echo "== setprio"; CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_prio.so timeout 600 python tools/bench_gemm.py time 2>&1 | grep -v amdgpu.ids | cut -c1-160
done
