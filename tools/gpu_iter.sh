#!/bin/bash
# iteration visit: gpu tests + kernel microbench + short bench
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -25 gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_kernels.py $MICRO > gpurun_out/micro.log 2>&1; cat gpurun_out/micro.log | grep -v amdgpu.ids
if [ -n "$MICRO2ENV" ]; then env $MICRO2ENV timeout 600 python tools/bench_kernels.py $MICRO2 > gpurun_out/micro2.log 2>&1; cat gpurun_out/micro2.log | grep -v amdgpu.ids; fi
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_iter.log 2>&1; tail -1 gpurun_out/bench_iter.log | cut -c1-150
