#!/bin/bash
# prefetcher parameter sweep (bench.py, 20 steps), two passes
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f %.3f' % (d['value'], d['ms_per_step']), '$*')"; }
for pass in 1 2; do
  run
  run --prefetch-wgs 256 --prefetch-min-mb 0.25
  run --prefetch-wgs 512 --prefetch-min-mb 0.25
  run --prefetch-wgs 256 --prefetch-min-mb 0.05
  run --prefetch-wgs 512 --prefetch-min-mb 0.05
  run --prefetch-wgs 1024 --prefetch-min-mb 0.25
done
