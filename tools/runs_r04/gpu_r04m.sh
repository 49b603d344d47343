#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-train-step --no-profile "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('steps/s', d['value'], 'ms/step', d['ms_per_step'], 'steady', c['steady_step_ms'], 'render', c['render_step_ms'])"; }
for i in 1 2; do
echo "== no prefetch"; run --no-weight-prefetch
echo "== prefetch 128 lag 2 min 1MB"; run
echo "== prefetch 128 lag 2 min 8MB"; run --prefetch-min-mb 8
echo "== prefetch 128 lag 1 min 1MB"; run --prefetch-lag 1
echo "== prefetch 64 lag 2 min 4MB"; run --prefetch-wgs 64 --prefetch-min-mb 4
done
