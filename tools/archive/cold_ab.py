#!/usr/bin/env python3
"""Cold (one-shot plx_filter) call time for a list of plx_tune settings: python3 tools/cold_ab.py k=v[,k=v] ..."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
n, d, ell = 1_000_000, 8, float(os.environ.get("ELL", "1.0"))
g = torch.Generator().manual_seed(1234)
ref = (torch.randn(n, d, generator=g) / ell).cuda()
v = torch.randn(n, 1, generator=g).cuda()
out = torch.empty_like(v)
taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
lat = plx.Lattice()
for setting in sys.argv[1:] or ["sort_points=1"]:
    pairs = [kv.split("=") for kv in setting.split(",")]
    for k, val in pairs:
        nv.check(nv.lib().plx_tune(k.encode(), int(val)), "plx_tune")
    for _ in range(3):
        lat.filter_once(v, ref, taps, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        lat.filter_once(v, ref, taps, out)
    torch.cuda.synchronize()
    print(f"{setting:40s} cold call {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  m={lat.m}")
