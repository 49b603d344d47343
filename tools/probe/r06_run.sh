#!/bin/bash
# scratch: run on the GPU box
mkdir -p gpurun_out
STAMP_CFG=-1 timeout 600 python tools/probe/gemm4w_stamp.py > gpurun_out/gemm_small_stamp.txt 2>&1
tail -6 gpurun_out/gemm_small_stamp.txt | cut -c1-230
