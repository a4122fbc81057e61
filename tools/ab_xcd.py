#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
for ell in (1.0, 0.6931, 0.25):
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    vals, scratch, out = lat.new_values(1), lat.new_values(1), torch.empty_like(v)
    base = None
    for rm in (0, 1, 0, 1):
        tune("xcd_remap", rm)
        lat.splat(v, vals)
        ts = min(timeit(lambda: lat.splat(v, vals)) for _ in range(3))
        tb = min(timeit(lambda: lat.blur(vals, scratch, vd=1)) for _ in range(3))
        tl = min(timeit(lambda: lat.slice(vals, out, vd=1)) for _ in range(3))
        ta = min(timeit(lambda: lat.apply(v, out)) for _ in range(3))
        res = lat.apply(v).clone(); base = res if base is None else base
        print(f"ell={ell} m={lat.m} xcd_remap={rm}: splat {ts:6.2f} blur {tb:6.2f} slice {tl:6.2f} apply {ta:6.2f} us equal={torch.equal(res, base)}", flush=True)
    lat.close()
tune("xcd_remap", 1)
