#!/bin/bash
mkdir -p gpurun_out
echo "# same box, alternating: libcd360_pre.so = this tree with csrc/gemm8p.hip of commit 9c5e82d (before the epilogue work of round 6's second half); columns: steps/s, ms per step over 20 steps, steady step ms, render step ms" > gpurun_out/epilogue_ab.txt
for rep in 1 2 3; do
  for lib in libcd360_pre.so libcd360_hip.so; do
  CD360_LIB=$PWD/custom-diffusion360_amd/lib/$lib python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['config'].get('steady_step_ms'), d['config'].get('render_step_ms'))" >> gpurun_out/epilogue_ab.txt
  done
done
cat gpurun_out/epilogue_ab.txt
