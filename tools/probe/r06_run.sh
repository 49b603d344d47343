#!/bin/bash
# scratch: run on the GPU box
mkdir -p gpurun_out
python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3
