"""Per-launch view of a rocprofv3 --kernel-trace run: mean duration per (kernel, grid size), i.e. per layer SHAPE (the
--stats summary merges the 64x64, 32x32 and 16x16 launches of one kernel).  Usage: python tools/per_dispatch.py <trace dir> [blocks]"""
import csv, glob, os, re, sys
from collections import defaultdict

src = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(lambda: [0, 0.0])
order = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("pmp::", "")
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
        key = (name[:90], grid)
        if key not in acc:
            order.append(key)
        a = acc[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
print("%-92s %8s %7s %10s %9s %6s" % ("kernel", "wgs", "calls", "avg_us", "total_ms", "pct"))
for key in sorted(acc, key=lambda k: -acc[k][1]):
    n, us = acc[key]
    print("%-92s %8d %7d %10.1f %9.2f %6.2f" % (key[0], key[1], n, us / n, us / 1e3, 100 * us / tot))
