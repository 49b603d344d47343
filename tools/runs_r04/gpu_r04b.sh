#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python tools/bench_gemm.py qcheck > gpurun_out/r04b_qcheck.log 2>&1; echo "qcheck exit $?"; grep -c "^ok" gpurun_out/r04b_qcheck.log; grep "FAIL\|Error\|error" gpurun_out/r04b_qcheck.log | head -20
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x -k "dedup or headline" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
timeout 600 python tools/bench_gemm.py qattn 2>&1 | grep -v amdgpu.ids | cut -c1-120
CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_r04a.so timeout 600 python tools/bench_gemm.py qattn 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
