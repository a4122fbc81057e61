#!/usr/bin/env python3
"""What a blur pass costs without its neighbour gathers / id loads (diagnostic blur_ablate switch)."""
# needs the diagnostics build: make -C simplex_gp_amd/csrc diag && PLX_LIBRARY=$PWD/simplex_gp_amd/libplx_diag.so python tools/ablate_blur.py
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
vd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, vd, generator=g).cuda()
tune("compact_nbr", 0)
for ell in (1.0, 0.6931, 0.25):
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    vals, scratch = lat.new_values(vd), lat.new_values(vd)
    lat.splat(v, vals)
    for ab in ((0, 1, 3) if vd == 1 else (0, 1)):
        tune("blur_ablate", ab, lat)
        ts = [timeit(lambda: lat.blur(vals, scratch, vd=vd)) for _ in range(3)]
        print(f"vd={vd} ell={ell} m={lat.m} blur_ablate={ab} (1=no gathers, 3=no gathers + no id loads): {min(ts):.2f} us per 9 launches = {min(ts)/9:.2f} us each", flush=True)
    tune("blur_ablate", 0, lat)
    lat.close()
tune("compact_nbr", 1)
