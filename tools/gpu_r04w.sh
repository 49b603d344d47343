#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_capture_gpu.py tests/test_kernels_gpu.py -x -q -k "capture or memo or rowdot" > gpurun_out/r04w_tests.log 2>&1; echo "tests exit $?"; tail -15 gpurun_out/r04w_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], 'render', c['render_step_ms'], {k: round(v,3) for k,v in c['kernel_ms_per_step'].items() if k in ('rowdot4','plucker_features','row_stats')})"
