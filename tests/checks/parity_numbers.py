import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from oracle import oracle
oracle.set_exact_mode(False)
rng = np.random.default_rng(0)
taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
worst = 0
for n, d, vd, ell in [(100000, 4, 1, 1.0), (100000, 4, 11, 0.25), (50000, 8, 1, 0.7), (50000, 8, 11, 1.0), (20000, 3, 198, 1.0), (30000, 18, 3, 1.0)]:
    x = (rng.standard_normal((n, d)) / ell).astype(np.float32)
    v = rng.standard_normal((n, vd)).astype(np.float32)
    want = oracle.filter(v, x, taps)
    got = plx.filter(torch.from_numpy(v).cuda(), torch.from_numpy(x).cuda(), torch.from_numpy(taps)).cpu().numpy()
    e = np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want)
    worst = max(worst, e)
    print(n, d, vd, ell, f"rel-L2 vs oracle (duplicate-free mode) {e:.2e}", flush=True)
print("worst", f"{worst:.2e}")
