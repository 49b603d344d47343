#!/bin/bash
# SQ / cache PMC passes over single-shape micro-benchmarks (one kernel shape per run so per-kernel averages are meaningful)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES TCC_HIT_sum TCC_MISS_sum"
for K in ${2:-conv1 attn1 nerf1}; do
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    rm -rf /tmp/pmc_$K_$i
    timeout 600 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_${K}_$i -o m -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py $K > $OUT/${K}_$i.log 2>&1
    CC=$(find /tmp/pmc_${K}_$i -name "*counter_collection.csv" | head -1)
    python - "$CC" "$K" <<'PY' >> $OUT/summary.txt
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if not any(s in n for s in ("conv_igemm", "attn_fwd", "nerf_fused")):
        continue
    a = agg[n[:60]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(sys.argv[2], k)
    for c, (n, v) in sorted(d.items()):
        print(f"   {c:34s} avg/dispatch {v / n:16.1f}  (n={n})")
PY
  done
done
cat $OUT/summary.txt
