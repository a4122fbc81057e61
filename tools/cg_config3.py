#!/usr/bin/env python3
"""BASELINE.json config 3: N=1e6, d=8, RBFLattice order 1, 50 CG iterations on
(s K + sigma^2 I) with right-hand side [y | 10 Rademacher probes] (vd = 11),
GPyTorch default hyper-parameters (lengthscale = outputscale = softplus(0))."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers

n, d, iters = 1_000_000, 8, 50
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = torch.randn(n, generator=g).cuda()
Z = (torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1).cuda()
rhs = torch.cat([y[:, None], Z], 1)
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
res = {}
with torch.no_grad():
    for trial in range(3):
        plx.lattice_cache().clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mm = model.khat_matmul(x)
        sol, info = solvers.batched_cg(mm, rhs, max_iter=iters, tol=0.0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res = {"config": "N=1e6 d=8 vd=11 lengthscale=0.6931, 50 CG iterations incl. 1 lattice build",
               "seconds": round(dt, 4), "mvms_per_s": round(iters / dt, 1), "m_vertices": list(plx.lattice_cache()._entries.values())[0][0].m,
               "final_rel_residual_max": float(info["residual"].max())}
print(json.dumps(res))
