#!/bin/bash
# Per-kernel time of ONE steady denoise step: kernel traces of bench runs with K1 and K2 timed steps (eager launches, same warmup) are
# differenced, so start-up work, the render step and the warmup cancel.  tools/gpu_steady_diff.sh TAG [K1 K2]
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-s}; K1=${2:-3}; K2=${3:-11}
OUT=$GRAFT_REPO_ROOT/gpurun_out/steady_$TAG
mkdir -p $OUT
cd /tmp
for K in $K1 $K2; do
  rm -rf /tmp/sd_$K
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/sd_$K -o b -- python $GRAFT_REPO_ROOT/bench.py --steps $K --warmup 2 --no-cpu-baseline --no-train-step --no-profile ${STEADY_NOGRAPH---no-graph} > $OUT/log_$K.txt 2>&1
  echo "K=$K exit $?"
done
python - $(find /tmp/sd_$K1 -name "*kernel_trace.csv" | head -1) $(find /tmp/sd_$K2 -name "*kernel_trace.csv" | head -1) $((K2-K1)) > $OUT/steady_step.csv <<'PY'
import csv, sys, re
from collections import defaultdict
def load(path):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:120]
        key = (name, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"))
        a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg
a, b, n = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows = []
for k in b:
    dc, dt = b[k][0] - a.get(k, [0, 0.0])[0], b[k][1] - a.get(k, [0, 0.0])[1]
    if dc > 0: rows.append((dt / n, dc / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"steady step: {tot / 1e3:.3f} ms of kernels, {sum(r[1] for r in rows):.1f} launches")
print("us_per_step,launches_per_step,avg_us,grid,kernel")
for us, c, (name, grid) in rows: print(f"{us:.1f},{c:.1f},{us / c:.1f},{grid},\"{name}\"")
PY
head -70 $OUT/steady_step.csv | cut -c1-220
