#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
for (n, d, ell) in [(200, 1, 0.7), (10000, 2, 1.0), (100000, 4, 1.0), (20000, 8, 1.0)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1); out = torch.empty_like(v)
    for rep in range(2):
        for val in (0, 1):
            tune("blur_small", val, lat)
            t = min(timeit(lambda: lat.apply(v, out), iters=100) for _ in range(3))
            if val == 0: base = out.clone()
            print(f"n={n} d={d} m={lat.m} blur_small={val}: apply {t:6.2f} us identical={torch.equal(out, base)}", flush=True)
    lat.close()
