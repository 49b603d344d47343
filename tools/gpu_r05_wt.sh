#!/bin/bash
# A/B of the write-through output stores (cd360_tuning.store_wt): per shape, then the whole step, alternating
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python tools/bench_gemm.py store_wt 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wt_shapes.log
for rep in 1 2; do
  for wt in 0 1; do
    CD360_STORE_WT=$wt timeout 600 python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('store_wt=$wt rep $rep: %.2f steps/s steady %.2f ms render %.2f ms' % (d['value'], c['steady_step_ms'], c['render_step_ms']))" | tee -a gpurun_out/wt_bench.log
  done
done
