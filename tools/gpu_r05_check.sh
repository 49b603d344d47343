#!/bin/bash
# round-5 check visit: the whole -m gpu suite with its printed margins (not -x: every failure is listed), then the driver-style bench line
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r05a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log
grep -a "passed\|failed\|FAILED\|ERROR\|pytest exit" gpurun_out/pytest_$TAG.log | tail -30
if [ -z "$NO_BENCH" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_20.log 2>&1; echo "bench20 exit $?"; tail -1 gpurun_out/bench_${TAG}_20.log | cut -c1-400
fi
