#!/usr/bin/env python3
"""How far reference quirk Q1 moves the reference's OWN output: the CPU hash table hashes a lookup with the capacity in
force before the insert that doubles the table (permutohedral.h:105 before h:61-63), so the one lookup per doubling probes
from a stale bucket and may miss an existing key -- a duplicate, orphaned vertex (m + 1) whose share of the splat never
reaches its neighbours, or one neighbour read as absent.  The oracle runs both ways (exact mode = the reference bit for
bit; clean mode = the duplicate-free lattice, which is what the reference's CUDA table -- fixed capacity, cu:61, no growth
-- and the HIP path build); this script prints rel-L2(exact, clean) over shapes, lengthscales and seeds.  CPU only.
    python tests/checks/quirk_survey.py [--quick]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle

RBF1 = np.array([0.34608543, 1.0, 0.34608543], np.float32)
shapes = [(16599, 17, 0.6931), (16599, 17, 1.0), (100000, 4, 0.25), (100000, 4, 1.0), (20000, 8, 0.6931), (20000, 8, 1.0),
          (45730, 9, 0.6931), (48827, 20, 0.6931), (10623, 18, 1.0), (5000, 3, 0.1), (200000, 8, 0.6931), (50000, 2, 0.05)]
seeds = (1234, 1, 2) if "--quick" not in sys.argv else (1234,)
worst, rows = 0.0, []
for n, d, ell in shapes:
    for seed in seeds:
        g = torch.Generator().manual_seed(seed)
        x = (torch.randn(n, d, generator=g) / ell).contiguous().numpy()
        v = torch.randn(n, 1, generator=g).numpy()
        oracle.set_exact_mode(True)
        exact, m_exact = oracle.filter(v, x, RBF1, return_m=True)
        oracle.set_exact_mode(False)
        clean, m_clean = oracle.filter(v, x, RBF1, return_m=True)
        oracle.set_exact_mode(True)
        q = float(np.linalg.norm(exact.astype(np.float64) - clean) / np.linalg.norm(clean.astype(np.float64)))
        worst = max(worst, q)
        rows.append({"n": n, "d": d, "ell": ell, "seed": seed, "m_reference": int(m_exact), "m_duplicate_free": int(m_clean),
                     "duplicates": int(m_exact - m_clean), "rel_l2_reference_vs_duplicate_free": q})
        print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"cases": len(rows), "worst": worst, "median": float(np.median([r["rel_l2_reference_vs_duplicate_free"] for r in rows])),
                  "above_1e-4": sum(r["rel_l2_reference_vs_duplicate_free"] > 1e-4 for r in rows)}))
