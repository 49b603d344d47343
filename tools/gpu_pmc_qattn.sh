#!/bin/bash
# SQ PMC passes over the fused q-projection + attention kernel (A3 at both pose levels, A2 at both levels: tools/bench_gemm.py qattn_one):
#   tools/gpu_pmc_qattn.sh TAG ["ENV=.. ENV=.." ...]     ("-" = defaults) -> gpurun_out/pmc_sq_TAG/qproj_attn_a3_a2.txt
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
: > $OUT/qproj_attn_a3_a2.txt
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA"
[ $# -eq 0 ] && set -- "-"
v=0
for V in "$@"; do
  v=$((v+1))
  if [ "$V" = "-" ]; then V=""; fi
  echo "== variant $v: [$V]" >> $OUT/qproj_attn_a3_a2.txt
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    rm -rf /tmp/pq_${v}_$i
    env $V timeout 300 rocprofv3 --pmc $P --output-format csv -d /tmp/pq_${v}_$i -o m -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py qattn_one > $OUT/v${v}_$i.log 2>&1
    CC=$(find /tmp/pq_${v}_$i -name "*counter_collection.csv" | head -1)
    python - "$CC" <<'PY' >> $OUT/qproj_attn_a3_a2.txt
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("   (no counters:", e, ")"); rows = []
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_mfma" not in n:
        continue
    a = agg[n[:78] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print("  ", k)
    for c, (n, v) in sorted(d.items()):
        print(f"      {c:34s} avg/dispatch {v / n:16.1f}  (n={n})")
PY
  done
done
cat $OUT/qproj_attn_a3_a2.txt
