"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid, workgroup): total / calls / average.  Usage: prof_summary.py trace.csv [topN]"""
import csv
import sys
from collections import defaultdict

path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
agg = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name")
        grid = r.get("Grid_Size_X", "") + "x" + r.get("Grid_Size_Y", "")
        wg = r.get("Workgroup_Size_X", "")
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = agg[(name, grid, wg)]
        a[0] += 1
        a[1] += d
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(v[0] for v in agg.values())} dispatches")
print("total_ms,pct,calls,avg_us,grid,wg,kernel")
for (name, grid, wg), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{d / 1e6:.3f},{100 * d / tot:.2f},{n},{d / n / 1e3:.1f},{grid},{wg},{name[:140]}")
