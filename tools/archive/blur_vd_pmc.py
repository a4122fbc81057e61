#!/usr/bin/env python3
"""Workload for a PMC pass over the multi-column blur: N=1e6, d=8, one lengthscale, one vd, 5 blurs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import RBF1
ell, vd = float(sys.argv[1]), int(sys.argv[2])
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
vals, scr = lat.new_values(vd), lat.new_values(vd)
for _ in range(5):
    vals.normal_()
    lat.blur(vals, scr, vd=vd)
torch.cuda.synchronize()
print("m", lat.m, "row bytes", lat.values_stride(vd) * 4)
