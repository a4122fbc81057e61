#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
vd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, vd, generator=g).cuda()
for ell in (1.0, 0.6931):
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    vals, scratch, out = lat.new_values(vd), lat.new_values(vd), torch.empty_like(v)
    base = None
    for rm in (0, 1, 0, 1):
        tune("xcd_remap", rm, lat)
        lat.splat(v, vals)
        ts = min(timeit(lambda: lat.splat(v, vals)) for _ in range(3))
        tb = min(timeit(lambda: lat.blur(vals, scratch, vd=vd)) for _ in range(3))
        tl = min(timeit(lambda: lat.slice(vals, out, vd=vd)) for _ in range(3))
        ta = min(timeit(lambda: lat.apply(v, out)) for _ in range(3))
        res = lat.apply(v).clone(); base = res if base is None else base
        print(f"ell={ell} vd={vd} m={lat.m} xcd_remap={rm}: splat {ts:7.2f} blur {tb:7.2f} slice {tl:7.2f} apply {ta:7.2f} us equal={torch.equal(res, base)}", flush=True)
    lat.close()
tune("xcd_remap", 1)
