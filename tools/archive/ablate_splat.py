#!/usr/bin/env python3
# needs the diagnostics build: make -C simplex_gp_amd/csrc diag && PLX_LIBRARY=$PWD/simplex_gp_amd/libplx_diag.so python tools/ablate_splat.py
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
for ell in (1.0, 0.25):
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    vals = lat.new_values(1)
    for ab in (0, 1, 2, 4, 3, 6, 7):
        tune("splat_ablate", ab, lat)
        ts = [timeit(lambda: lat.splat(v, vals)) for _ in range(3)]
        print(f"ell={ell} ablate={ab} (1=no gather 2=no stores 4=no rowid): {min(ts):.2f} us", flush=True)
    tune("splat_ablate", 0, lat)
    lat.close()
