#!/bin/bash
# scratch: run on the GPU box
mkdir -p gpurun_out
rm -f gpurun_out/prefetch_sweep.txt
for rep in 1 2; do
for cfg in "2 256 0.25" "2 128 0.25" "2 512 0.25" "2 1024 0.25" "2 256 0.1" "2 256 1" "3 256 0.25" "2 2048 0.25"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile --prefetch-lag $1 --prefetch-wgs $2 --prefetch-min-mb $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lag $1 wgs $2 min_mb $3:', d['value'], d['ms_per_step'], d['config'].get('steady_step_ms'))" >> gpurun_out/prefetch_sweep.txt
done
done
cat gpurun_out/prefetch_sweep.txt
