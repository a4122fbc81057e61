"""Round 6: steady-state time of one epoch of training.fit (step + evaluation of two splits) at any shape, and where it goes:
    python tools/loop_epoch_r6.py n d [--matern ORDER | --rbf ORDER] [--epochs E]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers, training                              # noqa: E402

n, d = int(sys.argv[1]), int(sys.argv[2])
order = int(sys.argv[sys.argv.index("--matern") + 1]) if "--matern" in sys.argv else int(sys.argv[sys.argv.index("--rbf") + 1]) if "--rbf" in sys.argv else 1
epochs = int(sys.argv[sys.argv.index("--epochs") + 1]) if "--epochs" in sys.argv else 12
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
X = torch.randn(n + n // 2, d, generator=g)
Y = torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1 % d]) + 0.1 * torch.randn(X.shape[0], generator=g)
X, Y = X.to(dev), Y.to(dev)
tr, va, te = slice(0, n), slice(n, n + n // 4), slice(n + n // 4, n + n // 2)
kern = plx.MaternLattice(nu=1.5, order=order, ard_num_dims=d) if "--matern" in sys.argv else plx.RBFLattice(order=order, ard_num_dims=d)
model = solvers.LatticeGP(kern, min_noise=1e-2).to(dev)
solvers.cap_host_threads()
stamps = []


def stamp(row):
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())


training.fit(model, (X[tr], Y[tr]), val=(X[va], Y[va]), test=(X[te], Y[te]), epochs=epochs, lr=0.1, num_probes=10, cg_iter=500,
             cg_tol=1.0, cg_eval_tol=1e-2, lanc_iter=100, pre_size=100, log=stamp)
per = [(b - a) * 1e3 for a, b in zip(stamps[2:-1], stamps[3:])]
print(f"n = {n}, d = {d}, {type(kern).__name__} order {order}: epoch (step + two splits) min {min(per):.1f} ms, median {sorted(per)[len(per) // 2]:.1f} ms", flush=True)
# where it goes, at the final hyper-parameters
prof = {}
opt = training.make_optimizer(model)
opt.zero_grad()
torch.cuda.synchronize()
t0 = time.perf_counter()
mll = solvers.marginal_log_likelihood(model, X[tr], Y[tr], num_probes=10, max_cg_iter=500, cg_tol=1.0, pre_size=100, profile=prof)
torch.cuda.synchronize()
t1 = time.perf_counter()
(-mll).backward()
torch.cuda.synchronize()
prof["backward"] = (time.perf_counter() - t1) * 1e3
print("   step phases (ms):", {k: round(v, 2) for k, v in prof.items()}, "cg iterations", mll.cg_info.get("iterations"), flush=True)
with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache = training.PredictionCache(model, X[tr], Y[tr], cg_tol=1e-2, lanc_iter=100, pre_size=100)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    cache.predict(X[va])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"   prediction cache {1e3 * (t1 - t0):.2f} ms (cg iterations {cache.solve_info.get('iterations')}), one split {1e3 * (t2 - t1):.2f} ms; "
      f"m = {[e[0].m for e in plx.lattice_cache()._entries.values()]}", flush=True)
