#!/usr/bin/env python3
"""Round 5: the pivoted-Cholesky factor's wall time against the speculation depth (candidates per batch) and the two
switches of its build (kernel rows on the frontier / by the dense stages; exact step launches behind the plan), at
config-3 size.  python tools/pchol_batch_sweep.py [n] [rank] [ell]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
with torch.no_grad():
    if len(sys.argv) > 3:
        model.kernel.lengthscale = float(sys.argv[3])
    lat = model.preconditioner(x, rank).lat
    s, noise = float(model.outputscale), float(model.noise)
    for kw in (dict(batch=12), dict(batch=8), dict(batch=16), dict(batch=12, sparse_rows=False), dict(batch=12, exact_steps=True),
               dict(batch=12, exact_steps=False), dict(batch=16, exact_steps=False), dict(batch=4)):
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pre = solvers.LatticePreconditioner(lat, s, noise, rank, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(json.dumps({"m": lat.m, **kw, "factor_ms": round(best * 1e3, 3), "batches": pre.batches, "planned_batches": pre.planned_batches,
                          "sparse_batches": pre.sparse_batches}), flush=True)
