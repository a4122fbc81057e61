#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1, tune
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
for ell in (1.0, 0.6931, 0.4, 0.25):
    base = None
    for mode in (0, 2):
        tune("compact_nbr", mode)
        lat = plx.Lattice(); lat.set_timing(True); lat.build((x / ell).contiguous().cuda(), RBF1); bt = lat.build_times_ms(); lat.set_timing(False)
        vals, scratch = lat.new_values(1), lat.new_values(1)
        lat.splat(v, vals)
        ts = [timeit(lambda: lat.blur(vals, scratch, vd=1)) for _ in range(3)]
        out = lat.apply(v).clone()
        base = out if base is None else base
        print(f"ell={ell} m={lat.m} compact={mode}: blur {min(ts):.2f} us ({min(ts)/9:.2f}/launch) nbr-stage build {bt['neighbours']:.3f} ms  equal={torch.equal(out, base)}", flush=True)
        lat.close()
tune("compact_nbr", 1)
