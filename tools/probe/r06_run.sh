#!/bin/bash
# scratch: run on the GPU box
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "skip_convolution or out_conv4 or halo" 2>&1 | tail -15 > gpurun_out/skip_tests.log
python -m pytest tests/test_modules_gpu.py tests/test_f_rows_gpu.py -q -x 2>&1 | tail -8 >> gpurun_out/skip_tests.log
for rep in 1 2; do
  for v in 0 1; do
    CD360_NO_SKIP_FUSE=$v python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no_skip_fuse=$v', d['value'], d['ms_per_step'], d.get('config',{}).get('steady_ms'))" >> gpurun_out/skip_ab.log
  done
done
cat gpurun_out/skip_tests.log gpurun_out/skip_ab.log
