#!/usr/bin/env python3
"""Round 6: the wide blur on sparse lattices -- only the rows with a neighbour on the pass' axis, in place
(plx_tune("blur_active", 1 / 2)) against the dense passes (0).  Whole blur of W columns, interleaved, minimum over rounds;
outputs compared.

    python tools/ab_blur_active_r6.py [rounds]
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cases = [(10623, 18, 1.0, 3, 418, "config-5 stand-in, backward width"), (16599, 17, 0.6931, 1, 396, "elevators shape, L = 11"),
         (48827, 20, 0.6931, 1, 462, "keggdirected shape, L = 11"), (45730, 9, 0.6931, 1, 220, "protein shape (43 % of the rows change)"),
         (1_000_000, 8, 0.25, 1, 198, "N = 1e6 fine regime"), (1_000_000, 8, 0.4, 1, 198, "N = 1e6, l = 0.4 (50 % change)")]


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for n, d, ell, order, W, label in cases:
    g = torch.Generator().manual_seed(1234)
    x = (torch.randn(n, d, generator=g) / ell).contiguous().cuda()
    taps = plx.DiscretizedKernelFN(plx.rbf if order == 1 else (lambda d2: plx.Matern.apply(d2, 1.5)), order).get_deriv_coeffs().numpy()
    lats, outs, kinds = {}, {}, {}
    for mode in (0, 2):
        nv.check(nv.lib().plx_tune(b"blur_active", mode), "tune")
        lats[mode] = plx.Lattice().build(x, taps)
    nv.check(nv.lib().plx_tune(b"blur_active", 1), "tune")
    m = lats[0].m
    vdp = lats[0].values_stride(W)
    vals = torch.randn(m, vdp, device="cuda")
    a, b = torch.empty_like(vals), torch.empty_like(vals)
    best = {0: 1e9, 2: 1e9}
    for r in range(rounds + 1):
        for mode, lat in lats.items():
            a.copy_(vals)
            t0 = sync()
            res = lat.blur(a, b, vd=W)
            t1 = sync()
            if r == 0:
                outs[mode] = res.clone()
                kinds[mode] = lat.stage_kernels()["blur_axis"]
            else:
                best[mode] = min(best[mode], (t1 - t0) * 1e3)
    nbr_share = None
    if m <= 2_000_000:
        nbr = lats[0].export(nv.ARRAY_NEIGHBORS)
        nbr_share = round(float((nbr >= 0).any(axis=1).mean()), 4)
    print(json.dumps({"case": label, "n": n, "d": d, "lengthscale": ell, "order": order, "columns": W, "m": m, "rows_changed_share": nbr_share,
                      "dense_ms": round(best[0], 3), "active_rows_ms": round(best[2], 3), "equal": bool(torch.equal(outs[0], outs[2])),
                      "kernels": kinds}), flush=True)
    for lat in lats.values():
        lat.close()
    del vals, a, b, outs
    torch.cuda.empty_cache()
