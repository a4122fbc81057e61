"""How sparse a blur axis is: the share of vertices with at least one existing neighbour on an axis (the vertices a pass changes
when the centre tap is 1), per lattice.  python tests/checks/active_fraction.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench
cases = [(10623, 18, 1.0, 3, "config 5 stand-in (Matern order 3)"), (10623, 18, 0.6931, 3, "same, lengthscale 0.69"), (16599, 17, 0.6931, 1, "elevators shape"),
         (48827, 20, 0.6931, 1, "keggdirected shape"), (45730, 9, 0.6931, 1, "protein shape"), (1_000_000, 8, 0.25, 1, "fine regime"),
         (1_000_000, 8, 0.4, 1, "l = 0.4"), (1_000_000, 8, 0.6931, 1, "config 3"), (200_000, 8, 0.3, 1, "N = 2e5, l = 0.3")]
for n, d, ell, order, label in cases:
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g)
    taps = plx.DiscretizedKernelFN(plx.rbf if order == 1 else (lambda d2: plx.Matern.apply(d2, 1.5)), order).get_coeffs().numpy()
    lat = plx.Lattice().build((x / ell).contiguous().cuda(), taps)
    nbr = lat.export(nv.ARRAY_NEIGHBORS)                       # [d+1, 2r, m]
    active = (nbr >= 0).any(axis=1)                            # [d+1, m]
    frac = active.mean(axis=1)
    print(f"{label}: n={n} d={d} l={ell} order={order} m={lat.m} m/nnz={lat.m / (n * (d + 1)):.3f} active share per axis: "
          f"min {frac.min():.4f} mean {frac.mean():.4f} max {frac.max():.4f}; slots present {float((nbr >= 0).mean()):.4f}", flush=True)
    lat.close()
