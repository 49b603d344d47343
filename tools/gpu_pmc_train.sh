#!/bin/bash
# HBM traffic of the fine-tune step's kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md;
# never combined with tracing).  Raw counter values: FETCH_SIZE is in units of 64 B on gfx950 and must be DOUBLED from what the
# tool prints as KiB-style units (see profiles/README.md); the summaries keep the raw numbers.  Outputs under gpurun_out/prof_$TAG.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-trainpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rm -rf /tmp/prof_$TAG
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$TAG/$C -o train -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 1 --warmup 1 > $OUT/train_$C.log 2>&1; echo "$C exit $?"
  CC=$(find /tmp/prof_$TAG/$C -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $CC $C attn_bwd nerf_bwd volrender_bwd gn_bwd add_layernorm_bwd geglu_bwd feature_gather conv_igemm attn_fwd attn_smallk nerf_fused adamw rowdot4 gemm_tn tn_reduce --by-grid > $OUT/pmc_$C.csv
  head -30 $OUT/pmc_$C.csv | cut -c1-160
done
