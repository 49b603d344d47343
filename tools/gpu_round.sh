#!/bin/bash
# One GPU-box visit: parity tests, smoke, short bench.  Everything is logged under gpurun_out/.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
echo "== rocm-smi ==" > gpurun_out/env.log; rocm-smi --showproductname 2>&1 | head -20 >> gpurun_out/env.log; nproc >> gpurun_out/env.log; free -g >> gpurun_out/env.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -8 gpurun_out/smoke.log
if [ "$1" == "bench" ]; then
  timeout 900 python bench.py --steps ${2:-10} --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -12 gpurun_out/bench.log
fi
