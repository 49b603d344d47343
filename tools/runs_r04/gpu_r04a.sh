#!/bin/bash
# round 4, visit a: parity of the restructured fused q-projection + attention epilogue, then same-box A/B against the round-3 library
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python tools/bench_gemm.py qcheck > gpurun_out/r04a_qcheck.log 2>&1; echo "qcheck exit $?" | tee -a gpurun_out/r04a_qcheck.log
grep -c "^ok" gpurun_out/r04a_qcheck.log; grep "FAIL\|Error\|error" gpurun_out/r04a_qcheck.log | head -20
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x -k "dedup or headline" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/bench_gemm.py qattn > gpurun_out/r04a_qattn_new.log 2>&1; grep -v amdgpu.ids gpurun_out/r04a_qattn_new.log
CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_r03.so timeout 600 python tools/bench_gemm.py qattn > gpurun_out/r04a_qattn_r03.log 2>&1; grep -v amdgpu.ids gpurun_out/r04a_qattn_r03.log
