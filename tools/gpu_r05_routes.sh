#!/bin/bash
# round 5: -m gpu suite, then the routes of one sampler side by side on one box (fused / sample.py-style rebinding / strict module route)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r05b}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log
grep -a "passed\|failed\|FAILED\|ERROR\|pytest exit" gpurun_out/pytest_$TAG.log | tail -30
for route in fused sample_py module; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile --route $route 2>gpurun_out/route_$route.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('route $route: %.2f steps/s steady %.2f ms render %.2f ms' % (d['value'], c['steady_step_ms'], c['render_step_ms']))" | tee -a gpurun_out/routes_$TAG.log
done
CD360_STRICT_SAMPLE_PY=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile --route sample_py 2>&1 | tail -3 | cut -c1-300 | tee -a gpurun_out/routes_$TAG.log
