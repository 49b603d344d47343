"""Per-kernel average of a rocprofv3 --pmc counter from counter_collection.csv.  Usage: pmc_summary.py file.csv COUNTER [substr ...]"""
import csv
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
want = [a for a in sys.argv[3:] if not a.startswith('--')]
agg = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != counter:
            continue
        name = r.get("Kernel_Name", "")
        if want and not any(w in name for w in want):
            continue
        if "--by-grid" in sys.argv:
            name = name[:60] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        a = agg[name[:100]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print("kernel,dispatches,avg_%s,total" % counter)
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"\"{k}\",{n},{v / n:.1f},{v:.1f}")
