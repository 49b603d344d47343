#!/bin/bash
# Per-kernel time of ONE fine-tune step (tools/bench_train.py, BASELINE config 4): kernel traces of runs with K1 and K2 timed steps are
# differenced, so model construction / random init / warmup cancel.  [TRAIN_ARGS=--graph] tools/gpu_steady_diff_train.sh TAG [K1 K2 [rows]]
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-t}; K1=${2:-1}; K2=${3:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/steady_train_$TAG
mkdir -p $OUT
cd /tmp
for K in $K1 $K2; do
  rm -rf /tmp/sdt_$K
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/sdt_$K -o b -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps $K --warmup 1 $TRAIN_ARGS > $OUT/log_$K.txt 2>&1
  echo "K=$K exit $?"
done
python - $(find /tmp/sdt_$K1 -name "*kernel_trace.csv" | head -1) $(find /tmp/sdt_$K2 -name "*kernel_trace.csv" | head -1) $((K2-K1)) > $OUT/steady_train_step.csv <<'PY'
import csv, sys, re
from collections import defaultdict
def load(path):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"at::native::", "", name)[:150]
        a = agg[(name, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"))]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg
a, b, n = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows = []
for k in b:
    dc, dt = b[k][0] - a.get(k, [0, 0.0])[0], b[k][1] - a.get(k, [0, 0.0])[1]
    if dc > 0: rows.append((dt / n, dc / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"fine-tune step: {tot / 1e3:.3f} ms of kernels, {sum(r[1] for r in rows):.1f} launches")
print("us_per_step,launches_per_step,avg_us,grid,kernel")
for us, c, (name, grid) in rows: print(f"{us:.1f},{c:.1f},{us / c:.1f},{grid},\"{name}\"")
PY
head -${4:-90} $OUT/steady_train_step.csv | cut -c1-200
