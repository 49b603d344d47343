#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -2 > gpurun_out/lbw_tests.txt
cat gpurun_out/lbw_tests.txt
rm -f gpurun_out/lbw_bench.txt
for rep in 1 2; do
  for lib in libcd360_old.so libcd360_lbw.so libcd360_hip.so; do
  CD360_LIB=$PWD/custom-diffusion360_amd/lib/$lib python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['config'].get('steady_step_ms'), d['config'].get('render_step_ms'))" >> gpurun_out/lbw_bench.txt
  done
done
cat gpurun_out/lbw_bench.txt
