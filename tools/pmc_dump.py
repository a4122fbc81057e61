#!/usr/bin/env python3
"""Per-kernel averages of the counters of a rocprofv3 --pmc ... --output-format csv run: python tools/pmc_dump.py <dir> [name filter]."""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    if flt in name:
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} {sum(v) / len(v):16.1f}   (n={len(v)})")
