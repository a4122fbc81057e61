#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
for vd in (64, 198, 418):
    v = torch.randn(n, vd, generator=g).cuda()
    for ell in (1.0, 0.6931):
        lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
        vals = lat.new_values(vd)
        base = None
        for mode in (0, 1, 0, 1):
            tune("splat_wide", mode, lat)
            ts = min(timeit(lambda: lat.splat(v, vals), iters=5) for _ in range(2))
            res = lat.splat(v, vals).clone(); base = res if base is None else base
            err = ((res - base).norm() / base.norm()).item()
            print(f"vd={vd} ell={ell} m={lat.m} splat_wide={mode}: splat {ts:9.1f} us  rel diff vs scan {err:.1e}", flush=True)
        lat.close()
    del v
