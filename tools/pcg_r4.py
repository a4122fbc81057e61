#!/usr/bin/env python3
"""Round 4: the preconditioned solve of the training recipe at config-3 size (N=1e6, d=8, l=0.6931, [y | 10 probes],
pre_size 100): factor construction, preconditioned vs plain CG iteration, and the recurrence residual against the
residual recomputed with the HIP operator."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = torch.randn(n, generator=g).cuda()
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    K = model.kernel(x, x)
    for rep in range(3):
        t0 = sync()
        pre = model.preconditioner(x, rank, K=K)
        t1 = sync()
        print(json.dumps({"factor_ms": round((t1 - t0) * 1e3, 2), "batches": pre.batches}), flush=True)
    Zp = pre.sample(10, generator=torch.Generator(device="cuda").manual_seed(0))
    rhs = torch.cat([y[:, None], Zp], 1).contiguous()
    mm = model.khat_matmul(x, K)
    for iters in (20, 50):
        for name, p in (("plain", None), ("pcg", pre)):
            for rep in range(3):
                t0 = sync()
                sol, info = model.khat_solve(x, rhs, K=K, max_iter=iters, tol=0.0, precond=p, want_tridiag=True)
                t1 = sync()
            true = ((mm(sol) - rhs).double().norm(dim=0) / rhs.double().norm(dim=0)).cpu()
            rep_ = info["residual"].double().cpu()
            print(json.dumps({"solve": name, "iters": iters, "ms": round((t1 - t0) * 1e3, 2), "us_per_iter": round((t1 - t0) / iters * 1e6, 1),
                              "reported_max": float(rep_.max()), "recomputed_max": float(true.max()),
                              "max_abs_diff": float((true - rep_).abs().max()), "max_rel_diff": float(((true - rep_) / true).abs().max())}), flush=True)
