#!/bin/bash
mkdir -p gpurun_out
python tools/probe/conv_halo_ab.py 2>&1 | grep "^conv" > gpurun_out/conv_halo_ab_fixed.txt; cat gpurun_out/conv_halo_ab_fixed.txt
python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3
rm -f gpurun_out/lbw_bench.txt
for rep in 1 2; do
  for lib in libcd360_old.so libcd360_hip.so; do
  CD360_LIB=$PWD/custom-diffusion360_amd/lib/$lib python bench.py --steps 20 --warmup 3 --no-train-step --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['config'].get('steady_step_ms'), d['config'].get('render_step_ms'))" >> gpurun_out/lbw_bench.txt
  done
done
cat gpurun_out/lbw_bench.txt
