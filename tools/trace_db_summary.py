#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (the default output when no --output-format is given)."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x from kernels").fetchall()
agg = collections.defaultdict(list)
for name, st, en, g in rows:
    nm = re.sub(r"\(.*", "", name.replace("void ", ""))[:78]
    agg[(nm, g)].append((en - st) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for (nm, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{nm:80s} grid {g:>10} calls {len(v):4d} mean {sum(v)/len(v):9.1f} us min {min(v):9.1f}")
