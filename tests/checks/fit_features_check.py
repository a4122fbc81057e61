#!/usr/bin/env python3
"""training.fit (step + evaluation of two splits per epoch) with round 6's sharing on -- one prediction cache per epoch, native
Lanczos step, lattice / preconditioner of unchanged hyper-parameters reused -- against the same loop with all of it off (every
evaluation from scratch, torch-op Lanczos, caches cleared between phases): the two histories must tell the same story (same
hyper-parameter trajectory to fp32 solver noise, RMSE / NLL within a few 1e-3), and the shared form must be the faster one.
    python tests/checks/fit_features_check.py [n] [d] [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers, training                              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 15
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
X = torch.randn(n + n // 2, d, generator=g)
f = torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) + 0.5 * X[:, 2 % d]
Y = f + 0.1 * torch.randn(X.shape[0], generator=g)
X, Y = X.to(dev), Y.to(dev)
tr, va, te = slice(0, n), slice(n, n + n // 4), slice(n + n // 4, n + n // 2)


def run(shared):
    torch.manual_seed(0)
    plx.lattice_cache().clear()
    model = solvers.LatticeGP(plx.MaternLattice(nu=1.5, order=2, ard_num_dims=d), min_noise=1e-2).to(dev)
    training.LANCZOS_NATIVE = shared
    real_cache = training.PredictionCache
    if not shared:
        class Fresh(real_cache):                           # nothing remembered: every cache starts from an empty lattice cache
            def __init__(self, m, *a, **k):
                plx.lattice_cache().clear()
                m.__dict__.pop("_last_preconditioner", None)
                super().__init__(m, *a, **k)
                m.__dict__.pop("_last_preconditioner", None)
        training.PredictionCache = Fresh
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist, best = training.fit(model, (X[tr], Y[tr]), val=(X[va], Y[va]), test=(X[te], Y[te]), epochs=epochs, lr=0.1,
                                  num_probes=10, cg_iter=500, cg_tol=1.0, cg_eval_tol=1e-2, lanc_iter=100, pre_size=100)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        training.PredictionCache = real_cache
        training.LANCZOS_NATIVE = True
    return hist, wall, {k: v.detach().flatten().cpu() for k, v in model.state_dict().items()}


hs, ws, ps = run(True)
hf, wf, pf = run(False)
print(f"shared: {ws / epochs * 1e3:.1f} ms per epoch; from scratch: {wf / epochs * 1e3:.1f} ms per epoch", flush=True)
worst = 0.0
for a, b in zip(hs, hf):
    for k in ("val/rmse", "test/rmse", "val/nll", "test/nll", "train/mll"):
        worst = max(worst, abs(a[k] - b[k]))
print("first epoch:", {k: round(v, 4) for k, v in hs[0].items()})
print("last epoch (shared):      ", {k: round(v, 4) for k, v in hs[-1].items()})
print("last epoch (from scratch):", {k: round(v, 4) for k, v in hf[-1].items()})
print(f"largest difference of any logged metric over the run: {worst:.2e}")
dp = max(float((ps[k] - pf[k]).abs().max()) for k in ps)
print(f"largest difference of any raw hyper-parameter at the end: {dp:.2e}")
assert hs[-1]["val/rmse"] < hs[0]["val/rmse"] and worst <= 2e-2 and dp <= 2e-2 and ws < wf
print("ok")
