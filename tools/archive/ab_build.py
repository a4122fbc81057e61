#!/usr/bin/env python3
"""Build-stage times for plx_tune variants (interleaved), N=1e6 d=8."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
from tools.archive.ab_apply import RBF1, tune
key = sys.argv[1] if len(sys.argv) > 1 else "insert_dedupe"
vals = [int(a) for a in sys.argv[2:]] or [0, 1]
n, d = 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g).cuda()
for ell in (1.0, 0.6931, 0.25):
    ref = (x / ell).contiguous().cuda()
    base = None
    for rep in range(2):
        for val in vals:
            tune(key, val)
            lat = plx.Lattice(); lat.set_timing(True)
            lat.build(ref, RBF1); lat.build(ref, RBF1)
            bt = lat.build_times_ms()
            keys = lat.export(nv.ARRAY_KEYS)
            out = lat.apply(v).clone()
            if base is None: base = (keys, out)
            same = np.array_equal(keys, base[0]) and torch.equal(out, base[1])
            print(f"ell={ell} {key}={val}: " + " ".join(f"{k}={t:.3f}" for k, t in bt.items()) + f" total={sum(bt.values()):.3f} ms  m={lat.m} identical={same}", flush=True)
            lat.close()
