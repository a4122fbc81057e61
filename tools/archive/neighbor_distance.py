#!/usr/bin/env python3
"""How far (in vertex ids) is a vertex from its blur neighbours, per axis?  Decides what the caches can hold."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
n, d = 1_000_000, 8
ell = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6931
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
lat = plx.Lattice().build((x / ell).contiguous().cuda(), taps)
nbr = lat.export(nv.ARRAY_NEIGHBORS)          # [d+1, 2, m]
m = lat.m
ids = np.arange(m)
print("m", m)
for j in range(d + 1):
    a = nbr[j, 1]
    ok = a >= 0
    dist = np.abs(a[ok] - ids[ok])
    print(f"axis {j}: present {ok.mean():.2f}  median {int(np.median(dist)):>8}  p90 {int(np.percentile(dist, 90)):>8}  "
          f"<=1k {np.mean(dist <= 1000):.2f}  <=10k {np.mean(dist <= 10000):.2f}  <=100k {np.mean(dist <= 100000):.2f}")
