#!/usr/bin/env python3
"""profiles/<tag>_pchol.md from a rocprofv3 rocpd database of tools/pchol_factor_timing.py: per-kernel means over the run and the
kernel timeline of the LAST factor build (one line per launch: start offset, gap to the previous kernel, duration).
    cd /tmp && rocprofv3 --kernel-trace --stats -d OUT -o pchol -- python3 tools/pchol_factor_timing.py
    python3 tools/make_pchol_profile.py <tag> OUT/pchol_results.db OUT/run.log"""
import collections, json, os, re, sqlite3, sys

tag, dbp, log = sys.argv[1:4]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
db = sqlite3.connect(dbp)
rows = db.execute("select name, start, end, grid_x from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", ""))[:70]
agg = collections.defaultdict(list)
for name, st, en, g in rows:
    agg[short(name)].append((en - st) / 1e3)
keep = [k for k in agg if k.startswith("plx::pchol") or k.startswith("plx::onehot") or "pcg_apply_kernel" in k or "pcg_to_half" in k or k.startswith("Cijk")]
lines = [f"| `{k}` | {len(agg[k])} | {sum(agg[k]) / len(agg[k]):.1f} | {min(agg[k]):.1f} |" for k in sorted(keep, key=lambda k: -sum(agg[k]))]
# the last factor build: from the last selection that follows a fill of the diagonal to the end
sel = [i for i, r in enumerate(rows) if "pchol_top_partial" in r[0]]
first = sel[-1]
while first > 0 and (rows[first][1] - rows[first - 1][2]) < 200_000 and not ("FillFunctor" in rows[first - 1][0] and rows[first - 1][3] > 500_000):
    first -= 1
seq = rows[first:]
t0 = seq[0][1]
tl, prev = [], None
for name, st, en, g in seq:
    gap = (st - prev) / 1e3 if prev else 0.0
    prev = en
    tl.append(f"{(st - t0) / 1e3:9.1f}  gap {gap:6.1f}  dur {(en - st) / 1e3:7.1f}  {short(name)}")
runs = [json.loads(l) for l in open(log) if l.startswith("{")]
md = f"""# rocprofv3 of the pivoted-Cholesky factor build ({tag})

`rocprofv3 --kernel-trace --stats -- python3 tools/pchol_factor_timing.py` (N = 1e6, d = 8, RBF order 1, GPyTorch's default
lengthscale 0.6931: m = 1.73e6; rank 100; `solvers.LatticePreconditioner`).  Wall times of the run (no profiler overhead
removed):

```
{chr(10).join(json.dumps(r) for r in runs[:8])}
```

| kernel | calls | mean us | min us |
|---|---:|---:|---:|
{chr(10).join(lines)}

Kernel timeline of the last build in the trace (us; under the profiler consecutive kernels abut, so `dur` includes the
launch turn-around; `gap` is GPU idle -- the host reading a batch's counts back):

```
{chr(10).join(tl[:140])}
```
"""
open(os.path.join(root, "profiles", f"{tag}_pchol.md"), "w").write(md)
print("wrote", f"profiles/{tag}_pchol.md", len(tl), "launches in the last build")
