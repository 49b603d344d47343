#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/probe/gemm4w_stamp.py > gpurun_out/gemm4w_stamp.txt 2>&1
tail -7 gpurun_out/gemm4w_stamp.txt | cut -c1-230
timeout 600 python tools/probe/gemm4w_ab.py time > gpurun_out/gemm4w_ab.txt 2>&1
tail -7 gpurun_out/gemm4w_ab.txt | cut -c1-200
