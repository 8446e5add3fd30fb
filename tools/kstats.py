#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats output directory (top N rows, short names)."""
import csv
import glob
import re
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
for f in files:
    rows = list(csv.DictReader(open(f)))
    print(f)
    for r in rows[:n]:
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("gvx::", "")
        print(f"  {name[:46]:46s} calls {int(r['Calls']):7d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {float(r['Percentage']):5.1f} %")
