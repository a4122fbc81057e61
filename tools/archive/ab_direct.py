#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
for (n, d, ell) in [(100000, 4, 1.0), (1000000, 8, 1.0), (1000000, 8, 0.6931), (1000000, 8, 0.25)]:
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    out = torch.empty_like(v); vals = lat.new_values(1)
    for rep in range(2):
        for val in (0, 1):
            tune("splat_direct", val, lat)
            ts = min(timeit(lambda: lat.splat(v, vals)) for _ in range(3))
            ta = min(timeit(lambda: lat.apply(v, out)) for _ in range(3))
            if val == 0: base = out.clone()
            print(f"n={n} d={d} ell={ell} splat_direct={val}: splat {ts:6.2f} apply {ta:6.2f} us identical={torch.equal(out, base)}", flush=True)
    lat.close()
