#!/usr/bin/env python3
"""Round 4: where one epoch of the config-5 stand-in goes (N=10,623, d=18, Matern-1.5 order 3, configs/simplexgp.yml recipe):
training step phases and the evaluation (training.predict) split into its parts."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers, training

n, d, ns = 10623, 18, 5976
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).cuda())
xs = torch.randn(ns, d, generator=g).cuda()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
model = solvers.LatticeGP(plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d), min_noise=0.1).cuda()
for rep in range(3):
    prof = {}
    t0 = sync()
    mll = solvers.marginal_log_likelihood(model, x, y, num_probes=10, cg_tol=1.0, max_cg_iter=500, seed=rep, pre_size=100, profile=prof)
    t1 = sync(); (-mll).backward(); t2 = sync()
    print(json.dumps({"train_forward_ms": round((t1 - t0) * 1e3, 2), "backward_ms": round((t2 - t1) * 1e3, 2),
                      "phases": {k: round(v, 2) for k, v in prof.items()}, "cg_iterations": mll.cg_info["iterations"]}), flush=True)
    model.zero_grad()
with torch.no_grad():
    for rep in range(3):
        t0 = sync()
        r = (y - model.mean).reshape(-1, 1)
        K = model.kernel(x, x)
        pre = model.preconditioner(x, 100, K=K); t1 = sync()
        alpha, info = model.khat_solve(x, r, K=K, max_iter=1000, tol=1e-2, precond=pre); t2 = sync()
        K_star = model.kernel(xs, x)
        mean = model.mean + model.outputscale * K_star.matmul(alpha).squeeze(-1); t3 = sync()
        Q, T = training.lanczos(model.khat_matmul(x), r.squeeze(-1), 100); t4 = sync()
        KQ = model.outputscale * K_star.matmul(Q.contiguous()); t5 = sync()
        print(json.dumps({"eval_ms": round((t5 - t0) * 1e3, 2), "preconditioner": round((t1 - t0) * 1e3, 2), "mean_solve": round((t2 - t1) * 1e3, 2),
                          "cg_iterations": info["iterations"], "rect_mvm_1col": round((t3 - t2) * 1e3, 2), "lanczos_100": round((t4 - t3) * 1e3, 2),
                          "rect_mvm_100col": round((t5 - t4) * 1e3, 2), "lanczos_steps": int(T.shape[0])}), flush=True)
    t0 = sync(); training.predict(model, x, y, xs, cg_tol=1e-2, lanc_iter=100, pre_size=100); t1 = sync()
    print(json.dumps({"predict_ms": round((t1 - t0) * 1e3, 2)}))
