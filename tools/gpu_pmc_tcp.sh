#!/bin/bash
# Vector-memory / L1 (TA, TCP) counters of one kernel: tools/gpu_pmc_tcp.sh TAG KERNEL_SUBSTR "command"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; FILTER=$2; CMD=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; rm -f $OUT/summary.txt
i=0
# (one validated counter set: a pass with a counter name this rocprofv3 does not know aborts and hangs until the timeout)
for P in "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/pt_$i
  (cd /tmp && timeout 120 rocprofv3 --pmc $P --output-format csv -d /tmp/pt_$i -o m -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/p$i.log 2>&1)
  CC=$(find /tmp/pt_$i -name "*counter_collection.csv" | head -1)
  python - "$CC" "$FILTER" <<'PY' >> $OUT/summary.txt
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
try: rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e: print("   (no counters:", e, ")"); rows = []
for r in rows:
    if sys.argv[2] not in r["Kernel_Name"]: continue
    a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for c, (n, v) in sorted(agg.items()): print(f"      {c:40s} avg/dispatch {v / n:18.1f}  (n={n})")
PY
done
cat $OUT/summary.txt; tail -3 $OUT/p1.log
