#!/bin/bash
# rocprofv3 passes over the bench workload: (1) kernel trace + stats, (2) PMC FETCH_SIZE, (3) PMC WRITE_SIZE (separate passes, as
# MI355X_MICROARCH.md prescribes; never combined with sys/runtime tracing).  Outputs under gpurun_out/prof_$TAG.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r1}
STEPS=${2:-6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-train-step --no-profile --no-graph"
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -o bench -- $BENCH > $OUT/bench_kt.log 2>&1; echo "kt exit $?"
find /tmp/prof_$TAG/kt -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
TRACE=$(find /tmp/prof_$TAG/kt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $TRACE 70 > $OUT/kernel_by_shape.csv
head -42 $OUT/kernel_by_shape.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$TAG/$C -o bench -- $BENCH > $OUT/bench_$C.log 2>&1; echo "$C exit $?"
  CC=$(find /tmp/prof_$TAG/$C -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $CC $C conv_igemm attn_fwd attn_self attn_smallk nerf_fused gn_ geglu volrender gemm_mfma row_stats > $OUT/pmc_$C.csv
  cat $OUT/pmc_$C.csv
done
cd $GRAFT_REPO_ROOT
if [ "$3" == "bench" ]; then timeout 900 python bench.py --steps 50 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_$TAG.log; fi
