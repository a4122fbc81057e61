#!/usr/bin/env python3
"""Rebuild + apply the same inputs many times and require bit-identical results (vertex count, outputs, gradients):
no kernel on the path may depend on scheduling (no float atomics; in-wave LDS hand-offs; CAS/atomicMin inserts
whose outcome is order-independent)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
g = torch.Generator().manual_seed(77)
bad = 0
for (n, d, vd, ell) in [(200000, 8, 1, 1.0), (200000, 8, 12, 0.7), (100000, 4, 40, 1.0), (50000, 8, 198, 0.7), (300000, 6, 1, 0.25), (100000, 3, 7, 0.2)]:
    x = (torch.randn(n, d, generator=g) / ell).cuda()
    v = torch.randn(n, vd, generator=g).cuda()
    lat = plx.Lattice()
    ref_out = ref_m = None
    for r in range(reps):
        lat.build(x, taps)
        out = lat.apply(v)
        if ref_out is None:
            ref_out, ref_m = out.clone(), lat.m
        elif lat.m != ref_m or not torch.equal(out, ref_out):
            bad += 1
            print("MISMATCH", (n, d, vd, ell), "rep", r, "m", lat.m, ref_m, float((out - ref_out).abs().max()), flush=True)
    lat.close()
    print((n, d, vd, ell), "m", ref_m, "identical over", reps, "rebuilds", flush=True)
# fused backward
n, d, L = 100000, 8, 11
x = (torch.randn(n, d, generator=g) / 0.7).cuda(); s = torch.randn(n, L, generator=g).cuda(); go = torch.randn(n, L, generator=g).cuda()
lat = plx.Lattice().build(x, taps)
a0 = b0 = None
for r in range(reps):
    a, b = lat.apply_backward(go, s, x)
    if a0 is None: a0, b0 = a.clone(), b.clone()
    elif not (torch.equal(a, a0) and torch.equal(b, b0)):
        bad += 1; print("MISMATCH backward rep", r, flush=True)
print("fused backward identical over", reps, "calls")
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
