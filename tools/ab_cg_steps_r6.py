#!/usr/bin/env python3
"""Round 6: the CG iteration with its column reductions inside the update / direction kernels (solvers.FUSED_CG_STEPS = True,
10 launches per iteration) against the one with them as stand-alone launches (False, 12 launches; round 5).  BASELINE.json
configs[2]: N = 1e6, d = 8, [y | 10 probes], 50 iterations on a cached lattice; plain and preconditioned (rank 100).
Interleaved in one process, minimum and median over rounds.

    python tools/ab_cg_steps_r6.py [rounds] [n] [d]
"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers

solvers.cap_host_threads()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = 50
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = torch.randn(n, generator=g).cuda()
Z = (torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1).cuda()
rhs = torch.cat([y[:, None], Z], 1)
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


with torch.no_grad():
    K = model.kernel(x, x)
    for ell in (0.6931, 0.8):
        model.kernel.lengthscale = ell
        K = model.kernel(x, x)
        pre = model.preconditioner(x, 100, K=K)
        for name, kw in (("plain", {}), ("pre_size_100", {"precond": pre})):
            t = {True: [], False: []}
            sols = {}
            for r in range(rounds + 1):
                for fused in (True, False):
                    solvers.FUSED_CG_STEPS = fused
                    t0 = sync()
                    sols[fused], info = model.khat_solve(x, rhs, K=K, max_iter=iters, tol=0.0, **kw)
                    t1 = sync()
                    if r > 0:
                        t[fused].append((t1 - t0) * 1e3)
            solvers.FUSED_CG_STEPS = "auto"
            rel = float((sols[True] - sols[False]).norm() / sols[False].norm())
            m = list(plx.lattice_cache()._entries.values())[-1][0].m
            print(json.dumps({"n": n, "d": d, "solve": name, "lengthscale": ell, "m": m, "iterations": iters,
                              "fused_ms": {"min": round(min(t[True]), 3), "median": round(float(np.median(t[True])), 3)},
                              "standalone_reductions_ms": {"min": round(min(t[False]), 3), "median": round(float(np.median(t[False])), 3)},
                              "solutions_rel_diff": rel}), flush=True)
