#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -s -k "ray_subset or configA or weight_cache or headline" > gpurun_out/r04c_new_tests.log 2>&1; echo "exit $?"
grep -v amdgpu.ids gpurun_out/r04c_new_tests.log | tail -60 | cut -c1-400
