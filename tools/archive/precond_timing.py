#!/usr/bin/env python3
"""Cost and benefit of the rank-k pivoted-Cholesky preconditioner at N=1e6, d=8 (config 3 operator)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).cuda())
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for raw_noise in (0.0, -4.0):
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        model.raw_noise.fill_(raw_noise)
        rhs = y[:, None].contiguous()
        for k in (0, 20, 100):
            for rep in range(2):
                t0 = sync()
                pre = model.preconditioner(x, k) if k else None
                t1 = sync()
                sol, info = model.khat_solve(x, rhs, max_iter=1000, tol=1e-3, precond=pre)
                t2 = sync()
            print(json.dumps({"noise": round(float(model.noise), 4), "pre_size": k, "build_ms": round((t1 - t0) * 1e3, 2),
                              "solve_ms": round((t2 - t1) * 1e3, 2), "iterations": info["iterations"]}), flush=True)
