#!/usr/bin/env python3
"""How many hashed inserts could a workgroup-level de-duplication save?  For consecutive points in lattice order:
corners that the wave-level rule of insert_kernel already skips (same vertex as the previous lane, same plane), and
distinct vertices per block of 256 / 512 points (what an LDS-level de-duplication would send to the global table)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
n, d = 1_000_000, 8
for ell in [float(a) for a in sys.argv[1:]] or [1.0, 0.6931, 0.25]:
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g)
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), np.array([0.34608543, 1.0, 0.34608543], np.float32))
    ev = lat.export(nv.ARRAY_ENTRY_VERTEX)            # [d+1, n] lattice order
    lead = 0
    for r in range(d + 1):
        v = ev[r]
        same = np.zeros(n, bool)
        same[1:] = v[1:] == v[:-1]
        same[::64] = False                             # lane 0 of every wave probes
        lead += int((~same).sum())
    out = [f"ell={ell} m={lat.m} corners={n * (d + 1)} wave-rule leaders={lead} ({lead / (n * (d + 1)):.2f})"]
    for B in (256, 512):
        nb = n // B
        blocks = ev[:, :nb * B].reshape(d + 1, nb, B).transpose(1, 0, 2).reshape(nb, -1)
        blocks.sort(axis=1)
        uniq = int((np.diff(blocks, axis=1) != 0).sum() + nb)
        out.append(f"distinct per {B}-point block: {uniq} ({uniq / (nb * B * (d + 1)):.2f})")
    print("; ".join(out), flush=True)
    lat.close()
