#!/bin/bash
# SQ PMC passes over one kernel: tools/gpu_pmc_kernel.sh TAG KERNEL_SUBSTR "python tools/bench_kernels.py attn2" "ENV=.. ENV=.." ...
# ("-" = no extra environment).  Counters only (no tracing domains), one pass per counter group.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; FILTER=$2; CMD=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
rm -f $OUT/summary.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"
P3="SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS"
v=0
for V in "$@"; do
  v=$((v+1))
  if [ "$V" = "-" ]; then V=""; fi
  echo "== variant $v: [$V]" >> $OUT/summary.txt
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rm -rf /tmp/pk_${v}_$i
    (cd /tmp && env $V timeout 300 rocprofv3 --pmc $P --output-format csv -d /tmp/pk_${v}_$i -o m -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/v${v}_$i.log 2>&1)
    CC=$(find /tmp/pk_${v}_$i -name "*counter_collection.csv" | head -1)
    python - "$CC" "$FILTER" <<'PY' >> $OUT/summary.txt
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("   (no counters:", e, ")"); rows = []
for r in rows:
    n = r["Kernel_Name"]
    if sys.argv[2] not in n:
        continue
    a = agg[n[:70] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print("  ", k)
    for c, (n, v) in sorted(d.items()):
        print(f"      {c:34s} avg/dispatch {v / n:16.1f}  (n={n})")
PY
  done
done
cat $OUT/summary.txt
