#!/bin/bash
# Build probe variants of the GEMM core with -DCD360_GEMM_SCHED=<bits> (gemm8p.hip: 1 s_setprio around the MFMA runs, 2 static priority for
# the second half of the workgroup, 4 fragment reads interleaved with the MFMAs by sched_group_barrier) as libcd360_sched<bits>.so.  Run HERE.
cd "$(dirname "$0")/../../custom-diffusion360_amd"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DCD360_GEMM_SCHED=$d -c csrc/gemm8p.hip -o /tmp/gemm8p_s$d.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libcd360_sched$d.so $(ls lib/obj/*.o | grep -v gemm8p) /tmp/gemm8p_s$d.o && echo built sched$d
done
