#!/bin/bash
# the driver's round-end sequence: every -m gpu test, smoke(), the default bench line
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/full_pytest.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/full_pytest.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/full_bench.log 2>&1; tail -1 gpurun_out/full_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('steps/s', d['value'], 'ms/step', d['ms_per_step'], 'steady', c['steady_step_ms'], 'render', c['render_step_ms'], 'prepare', c['prepare_ms'])
print({k: v for k, v in c['kernel_ms_per_step'].items() if v > 0.05})
print({k: (v['frac'], v.get('avg_us')) for k, v in d['rooflines'].items()})
print('train', {k: d['train_step'].get(k) for k in ('ms','library_graph_ms','cd360_ms','library_ms','error')})
print('cpu', d.get('cpu_baseline'))"
