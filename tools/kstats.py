#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --kernel-trace --stats --output-format csv run: tools/kstats.py <dir> [rows] [filter]"""
import csv, glob, os, sys
d = sys.argv[1]
rows_n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
filt = sys.argv[3] if len(sys.argv) > 3 else ""
path = max(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)     # the newest run of the directory
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{path}: {len(rows)} kernels, {total / 1e6:.2f} ms of kernel time")
for r in [r for r in rows if filt in r["Name"]][:rows_n]:
    name = r["Name"].replace("void ", "").split("(")[0][:78]
    print(f"{name:78s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms  mean {float(r['AverageNs']) / 1e3:9.2f} us  "
          f"min {float(r['MinNs']) / 1e3:9.2f}  {float(r['Percentage']):5.2f} %")
