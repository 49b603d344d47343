#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
timeout 400 python tools/bench_train.py --graph --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
CD360_TRAIN_NO_PREFETCH=1 timeout 400 python tools/bench_train.py --graph --steps 6 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
done
timeout 900 python -m pytest tests/test_modules_gpu.py -q -x -p no:cacheprovider -k "config4 or allreduce" 2>&1 | tail -2
