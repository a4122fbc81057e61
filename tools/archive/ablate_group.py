# needs the diagnostics build: make -C simplex_gp_amd/csrc diag && PLX_LIBRARY=$PWD/simplex_gp_amd/libplx_diag.so python tools/ablate_group.py
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
for ell in (1.0, 0.6931):
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    lat.set_lattice_row_order(True)
    for vd in (4, 11):
        v = torch.randn(n, vd, generator=g).cuda()
        vals = lat.new_values(vd)
        for ab in (0, 1, 2, 3):
            tune("splat_ablate", ab, lat)
            ts = min(timeit(lambda: lat.splat(v, vals), iters=5) for _ in range(3))
            print(f"ell={ell} vd={vd} ablate={ab} (1: no gathers, 2: no stores): splat {ts:8.1f} us", flush=True)
        tune("splat_ablate", 0, lat)
    lat.close()
