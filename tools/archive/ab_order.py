#!/usr/bin/env python3
"""A/B of the point order (plx_tune order_zcurve: 0 lexicographic, 1 Z-curve): neighbour distances and stage times."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
from tools.archive.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
vs = {vd: torch.randn(n, vd, generator=g).cuda() for vd in (1, 11, 198)}
for ell in (1.0, 0.6931):
    ref = (x / ell).contiguous().cuda()
    for basis in (0, 1):
        tune("order_zcurve", basis)
        lat = plx.Lattice().build(ref, RBF1)
        tb = min(timeit(lambda: lat.build(ref, RBF1), iters=3) for _ in range(2))
        nbr = lat.export(nv.ARRAY_NEIGHBORS)
        ids = np.arange(lat.m)
        med = []
        for j in range(d + 1):
            a = nbr[j, 1]; ok = a >= 0
            med.append(int(np.median(np.abs(a[ok] - ids[ok]))))
        line = f"ell={ell} zcurve={basis} m={lat.m} build {tb:8.1f} us  median nbr distance per axis {med}"
        print(line, flush=True)
        for vd, v in vs.items():
            if vd == 198 and ell == 0.25:
                continue
            vals, scr = lat.new_values(vd), lat.new_values(vd)
            out = torch.empty(n, vd, device="cuda")
            ts = min(timeit(lambda: lat.splat(v, vals), iters=5) for _ in range(2))
            tb_ = min(timeit(lambda: lat.blur(vals, scr, vd=vd), iters=5) for _ in range(2))
            res = lat.blur(vals, scr, vd=vd)
            tl = min(timeit(lambda: lat.slice(res, vd=vd, out=out), iters=5) for _ in range(2))
            ta = min(timeit(lambda: lat.apply(v, out), iters=5) for _ in range(2))
            print(f"    vd={vd:3d}: splat {ts:8.1f}  blur {tb_:8.1f}  slice {tl:8.1f}  apply {ta:8.1f} us", flush=True)
            del vals, scr, out, res
        lat.close()
