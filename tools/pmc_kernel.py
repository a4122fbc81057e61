#!/usr/bin/env python3
"""Mean of one PMC counter per kernel from a rocprofv3 --pmc ... --output-format csv directory."""
import collections, csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:62s} {c:12s} calls {len(v):4d} mean {sum(v)/len(v):14.1f}")
