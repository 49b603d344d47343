#!/bin/bash
# Per-kernel time of ONE render step (the first denoise step of a pose: FeatureNeRF renders of the 12 pose blocks + everything a steady step
# does): kernel traces of eager bench runs with 1 and 2 poses (K steps each) are differenced -- the difference is one render step + K - 1
# steady steps -- and K - 1 times the steady step (runs with K and K + 8 steps, as tools/gpu_steady_diff.sh) is taken off.
# tools/gpu_render_diff.sh TAG [K]
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r}; K=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/render_$TAG
mkdir -p $OUT
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --warmup 0 --no-cpu-baseline --no-train-step --no-profile --no-graph"
run() { rm -rf /tmp/rd_$1; shift_name=$1; shift; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/rd_$shift_name -o b -- $B "$@" > $OUT/log_$shift_name.txt 2>&1; echo "$shift_name exit $?"; }
run p1 --steps $K --poses 1
run p2 --steps $K --poses 2
run s2 --steps $((K + 8)) --poses 1
python - $(find /tmp/rd_p1 -name "*kernel_trace.csv" | head -1) $(find /tmp/rd_p2 -name "*kernel_trace.csv" | head -1) $(find /tmp/rd_s2 -name "*kernel_trace.csv" | head -1) $K > $OUT/render_step.csv <<'PY'
import csv, sys, re
from collections import defaultdict
def load(path):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"at::native::", "", name)[:120]
        a = agg[(name, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"))]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg
p1, p2, s2, K = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
keys = set(p1) | set(p2) | set(s2)
rows = []
for k in keys:
    g = lambda d: d.get(k, [0, 0.0])
    steady_c, steady_t = (g(s2)[0] - g(p1)[0]) / 8.0, (g(s2)[1] - g(p1)[1]) / 8.0
    c, t = g(p2)[0] - g(p1)[0] - (K - 1) * steady_c, g(p2)[1] - g(p1)[1] - (K - 1) * steady_t
    if c > 0.5 or t > 5.0: rows.append((t, c, k))
rows.sort(reverse=True)
print(f"render step: {sum(r[0] for r in rows) / 1e3:.3f} ms of kernels, {sum(r[1] for r in rows):.1f} launches")
print("us_per_step,launches_per_step,avg_us,grid,kernel")
for us, c, (name, grid) in rows: print(f"{us:.1f},{c:.1f},{us / max(c, 1e-9):.1f},{grid},\"{name}\"")
PY
head -60 $OUT/render_step.csv | cut -c1-200
