#!/bin/bash
# Probe build of the library with -DCD360_WHATIF: the GEMM core's what-if timing bits (cd360_tuning.whatif: results are WRONG when set)
# and the LDS-DMA variant of the render kernel (cd360_tuning.nerf_kernel = 2) are compiled in.  Links
# custom-diffusion360_amd/lib/libcd360_whatif.so; select it with CD360_LIB=<path> (cd360/_lib.py).  The product build has none of this code.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
P=$ROOT/custom-diffusion360_amd
[ -d $P/lib/obj ] || (cd $ROOT && python -c "import __graft_entry__ as g; g.build()")
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DCD360_WHATIF"
mkdir -p /tmp/cd360_whatif
for b in gemm8p nerf_fused tuning; do /opt/rocm/bin/hipcc $FLAGS -c $P/csrc/$b.hip -o /tmp/cd360_whatif/$b.o & done; wait
OBJS=$(for f in $P/csrc/*.hip; do b=$(basename $f .hip); case $b in gemm8p|nerf_fused|tuning) echo /tmp/cd360_whatif/$b.o;; *) echo $P/lib/obj/$b.o;; esac; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/libcd360_whatif.so $OBJS
echo built $P/lib/libcd360_whatif.so
