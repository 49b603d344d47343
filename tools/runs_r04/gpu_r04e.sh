#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -s -k "ray_subset or tolerance_report or configA" > gpurun_out/r04e_tests.log 2>&1; echo "exit $?"
grep "cfg-B level\|margins\|^  \|worst plain\|passed\|failed\|Error" gpurun_out/r04e_tests.log | cut -c1-330
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step > gpurun_out/r04e_bench.log 2>&1; tail -1 gpurun_out/r04e_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('bf16: steps/s', d['value'], 'steady', c['steady_step_ms'], 'render', c['render_step_ms'], 'prepare', c['prepare_ms'])
print({k: v for k, v in c['kernel_ms_per_step'].items() if v > 0.05})
print({k: (v['frac'], v.get('avg_us')) for k, v in d['rooflines'].items()})"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step --fp8-attn > gpurun_out/r04e_bench_fp8.log 2>&1; tail -1 gpurun_out/r04e_bench_fp8.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('fp8: steps/s', d['value'], 'steady', c['steady_step_ms'], 'render', c['render_step_ms'], c['attention_arith'])
print({k: v for k, v in c['kernel_ms_per_step'].items() if v > 0.05})
print(d['fp8_tolerance'])"
