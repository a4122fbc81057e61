#!/usr/bin/env python3
"""Multi-column blur: time per pass with the neighbour gathers on (0) / replaced by the centre row (1), with and
without the XCD tile remap.  Shows how much of a pass is the stream (centre read + write) and how much the gathers."""
# needs the diagnostics build: make -C simplex_gp_amd/csrc diag && PLX_LIBRARY=$PWD/simplex_gp_amd/libplx_diag.so python tools/ablate_blur_vd.py
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
for ell in (1.0, 0.6931, 0.25):
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
    for vd in (2, 4, 8, 11, 16):
        vals, scr = lat.new_values(vd), lat.new_values(vd)
        vals.normal_()
        ref = None
        for multi in (0, 1):
            tune("blur_narrow", multi, lat)
            vals.normal_()
            t = min(timeit(lambda: lat.blur(vals, scr, vd=vd), iters=5) for _ in range(3))
            res = lat.blur(vals, scr, vd=vd).clone()
            ref = res if ref is None else ref
            gb = lat.m * lat.values_stride(vd) * 4 * 2 * 9 / 1e3
            print(f"ell={ell} m={lat.m} vd={vd} narrow={multi}: blur {t:8.1f} us  ({t/9:6.1f}/pass)  stream {gb/t/1e3:.2f} TB/s  "
                  f"max diff vs single {float((res - ref).abs().max()):.1e}", flush=True)
        tune("blur_narrow", 1, lat)
    lat.close()
