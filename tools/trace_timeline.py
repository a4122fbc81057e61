#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 --kernel-trace --output-format csv run: start offset, gap, duration, name."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows[:limit]:
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:64]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name}")
    prev_end = e
