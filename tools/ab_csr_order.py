#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
for ell in (1.0, 0.6931):
    for pm in (0, 1, 0, 1):
        tune("csr_point_major", pm)
        lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
        vals = lat.new_values(1)
        ts = [timeit(lambda: lat.splat(v, vals)) for _ in range(3)]
        print(f"ell={ell} csr_point_major={pm}: splat {min(ts):.2f} us", flush=True)
        lat.close()
