#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for PPR in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step --poses $PPR --poses-per-replay $PPR > gpurun_out/r04k_ppr$PPR.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r04k_ppr$PPR.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('ppr', c['poses_per_replay'], 'pose-steps/s', d['value'], 'ms/replay', d['ms_per_step'], 'steady', c['steady_step_ms'], 'render', c['render_step_ms'])
print({k: v for k, v in c['kernel_ms_per_step'].items() if v > 0.1})
print({k: v['frac'] for k, v in d['rooflines'].items()})" || tail -5 gpurun_out/r04k_ppr$PPR.log
done
