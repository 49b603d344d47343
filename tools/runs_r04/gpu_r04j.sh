#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/bench_gemm.py check 2>&1 | grep -v amdgpu.ids | grep -c "^ok"; timeout 900 python tools/bench_gemm.py check 2>&1 | grep "FAIL" | head
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "conv" 2>&1 | tail -2
for i in 1 2; do
echo "== spread2 (new)"; timeout 600 python tools/bench_gemm.py time 2>&1 | grep -v amdgpu.ids | cut -c1-160
echo "== round-3 two-buffer loop"; CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_nospread2.so timeout 600 python tools/bench_gemm.py time 2>&1 | grep -v amdgpu.ids | cut -c1-160
done
