#!/bin/bash
# Probe build of the GEMM core with time stamps (-DCD360_GEMM_STAMP=MODE; MODE 1 = s_memtime stamps of every K-tile of the compiled loop,
# for gemm_stamp.py; MODE 2 (default) = 100 MHz phase stamps of a launch, for gemm4w_stamp.py): links custom-diffusion360_amd/lib/libcd360_stamp.so
# from the product objects plus the instrumented gemm8p.  Run here (cross-compiles); tools/probe/gemm_stamp.py uses it on the GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
P=$ROOT/custom-diffusion360_amd
[ -d $P/lib/obj ] || (cd $ROOT && python -c "import __graft_entry__ as g; g.build()")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DCD360_GEMM_STAMP=${1:-2} -c $P/csrc/gemm8p.hip -o /tmp/gemm8p_stamp.o
OBJS=$(for f in $P/csrc/*.hip; do b=$(basename $f .hip); [ $b = gemm8p ] || echo $P/lib/obj/$b.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/libcd360_stamp.so $OBJS /tmp/gemm8p_stamp.o
echo built $P/lib/libcd360_stamp.so
