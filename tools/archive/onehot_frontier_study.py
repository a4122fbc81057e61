#!/usr/bin/env python3
"""Round 5 study: how sparse is one kernel row K e_p (= slice(blur(splat(e_p))))?  For a few random points of a lattice:
the number of vertices with a non-zero value after every blur axis (the frontier), and the number of points the slice
then touches.  Host arithmetic on the exported tables.

    python tools/onehot_frontier_study.py [n] [d] [ell] [order]
"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ell = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6931
order = int(sys.argv[4]) if len(sys.argv) > 4 else 1
x, _ = bench.synth(n, d, 1)
lat = plx.Lattice()
half = np.linspace(0.1, 0.8, order, dtype=np.float32)          # tap VALUES do not matter here, only the stencil's reach
stencil = bench.RBF1 if order == 1 else np.concatenate([half, [1.0], half[::-1]]).astype(np.float32)
lat.build((x / ell).contiguous().cuda(), stencil)
m = lat.m
evid = lat.export(nv.ARRAY_ENTRY_VERTEX)          # [d+1][n]
nbr = lat.export(nv.ARRAY_NEIGHBORS)              # [d+1][2r][m]
r2 = nbr.shape[1]
rng = np.random.default_rng(0)
sizes, touched = [], []
is_front = np.zeros(m, dtype=bool)
for p in rng.integers(0, n, 24):
    front = np.unique(evid[:, p])
    per_axis = [int(front.size)]
    for a in range(d + 1):
        nb = nbr[a][:, front].reshape(-1)
        front = np.unique(np.concatenate([front, nb[nb >= 0]]))
        per_axis.append(int(front.size))
    sizes.append(per_axis)
    is_front[:] = False
    is_front[front] = True
    touched.append(int(is_front[evid].any(axis=0).sum()))
sizes = np.array(sizes)
print(json.dumps({"n": n, "d": d, "ell": ell, "order": order, "m": int(m),
                  "frontier_median_per_axis": np.median(sizes, 0).astype(int).tolist(),
                  "frontier_max_per_axis": sizes.max(0).tolist(),
                  "points_touched_median": int(np.median(touched)), "points_touched_max": int(max(touched))}), flush=True)
