#!/bin/bash
# fine-tune step after the torch-tail changes: gradient tests + the graphed step, A/B against the op-by-op torch forms
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_modules_gpu.py tests/test_linear_gpu.py -x -q -k "grad or train or backward or config4 or trainkeys or block or loss or live or finetune or adamw" > gpurun_out/r04r_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r04r_tests.log
for i in 1 2; do
  timeout 600 python tools/bench_train.py --steps 8 --warmup 2 --graph 2>&1 | tail -1 | cut -c1-200
  CD360_NO_TRAIN_FUSIONS=1 timeout 600 python tools/bench_train.py --steps 8 --warmup 2 --graph 2>&1 | tail -1 | cut -c1-200
done
