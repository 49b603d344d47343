#!/bin/bash
# block-level hooks keep fused internals: test + the three routes of bench.py
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_modules_gpu.py -x -q -k "hook or rebound or harvest or linear" > gpurun_out/r04s_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/r04s_tests.log
for r in fused hooked module; do
  timeout 600 python bench.py --steps 20 --warmup 5 --route $r --no-train-step --no-cpu-baseline > gpurun_out/r04s_$r.log 2>&1; echo "$r exit $?"
  tail -1 gpurun_out/r04s_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_step_ms'), d['config'].get('route'))"
done
