#!/bin/bash
# same-box A/B of bench.py under environment-variable variants: tools/gpu_ab.sh "VAR=a" "VAR=b OTHER=c" ...  ("-" = defaults)
export HSA_ENABLE_IPC_MODE_LEGACY=0
for V in "$@"; do
  if [ "$V" = "-" ]; then V=""; fi
  echo "== variant: [$V]"
  env $V python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('steps/s', d['value'], 'steady_ms', d['config']['steady_step_ms'], 'render_ms', d['config']['render_step_ms'])
print({k: v for k, v in d['config']['kernel_ms_per_step'].items() if v > 0.1})
print('roofline', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['avg_us'])"
done
