#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "attn or attention" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python tools/bench_attn_self.py 2>&1 | grep -v amdgpu.ids
CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_r04a.so timeout 300 python tools/bench_attn_self.py 2>&1 | grep -v amdgpu.ids
done
