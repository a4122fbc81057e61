#!/usr/bin/env python3
"""How wide are the index windows a splat chunk (1024 corners sorted by vertex -> point ids) and a slice tile
(256 points -> vertex ids) touch?  Decides whether an LDS-staged window could replace the L2 gathers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
n, d = 1_000_000, 8
ell = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
lat = plx.Lattice().build((x / ell).contiguous().cuda(), taps)
pt = lat.export(nv.ARRAY_CSR_POINT) & 0x7FFFFFFF
nn = (pt.size // 1024) * 1024
c = pt[:nn].reshape(-1, 1024)
span = c.max(1) - c.min(1)
print("m", lat.m, "splat chunk point-id span: median", int(np.median(span)), "p90", int(np.percentile(span, 90)), "max", int(span.max()))
for w in (4096, 16384, 32768):
    lo = np.median(c, axis=1, keepdims=True) - w // 2
    inside = ((c >= lo) & (c < lo + w)).mean()
    print(f"   fraction of corners inside a {w}-point window centred on the chunk median: {inside:.3f}")
ev = lat.export(nv.ARRAY_ENTRY_VERTEX)          # [d+1, n] in lattice order
nt = (n // 256) * 256
e = ev[:, :nt].reshape(d + 1, -1, 256).transpose(1, 0, 2).reshape(-1, (d + 1) * 256)
span = e.max(1) - e.min(1)
print("slice tile vertex-id span: median", int(np.median(span)), "p90", int(np.percentile(span, 90)), "max", int(span.max()))
for w in (4096, 16384, 32768):
    lo = np.median(e, axis=1, keepdims=True) - w // 2
    inside = ((e >= lo) & (e < lo + w)).mean()
    print(f"   fraction of corner reads inside a {w}-vertex window centred on the tile median: {inside:.3f}")
