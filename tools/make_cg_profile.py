#!/usr/bin/env python3
"""profiles/<tag>_cg.md from the outputs of tools/prof_cg_stats.sh <tag>_cg and tools/pmc_cg.sh (run by run_profiles.sh)."""
import os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
stats = open(os.path.join(out, f"{tag}_cg.txt")).read().splitlines()[:22]
pmc = open(os.path.join(out, f"{tag}_cg_pmc.txt")).read().rstrip()
md = f"""# rocprofv3 of tools/prof_cg.py ({tag})

BASELINE.json configs[2]: N = 1e6, d = 8, vd = 11 (12 columns), lengthscale 0.6931, 3 x (one lattice build + 50 CG iterations).
`tools/prof_cg_stats.sh {tag}_cg` (kernel trace + stats) and `tools/pmc_cg.sh` (one counter per pass, kernel trace only).

```
{chr(10).join(stats)}
```

PMC, mean per launch, counter unit KB (FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950: the two
cg_step kernels move 192 + 96 MB and 96 + 48 MB and read back 94 / 47 MB):

```
{pmc}
```
"""
open(os.path.join(root, "profiles", f"{tag}_cg.md"), "w").write(md)
print("wrote", f"profiles/{tag}_cg.md")
