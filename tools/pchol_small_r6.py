"""Round 6: the rank-100 pivoted-Cholesky factor at the configs[4] stand-in (N = 10,623, d = 18, Matern order 3) by itself:
wall time per build, batches, and (under rocprofv3) its kernels.  python tools/pchol_small_r6.py [builds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers                                        # noqa: E402

n, d = 10623, 18
builds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).to(dev)
model = solvers.LatticeGP(plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d), min_noise=0.1).to(dev)
solvers.cap_host_threads()
with torch.no_grad():
    pre = model.preconditioner(x, 100)
    s, noise = float(model.outputscale), float(model.noise)
    ts = []
    for _ in range(builds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p2 = solvers.LatticePreconditioner(pre.lat, s, noise, 100)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
print(f"factor build: min {min(ts):.2f} ms, median {sorted(ts)[len(ts) // 2]:.2f} ms; batches {p2.batches}, planned {p2.planned_batches}, "
      f"frontier {p2.sparse_batches}, m = {p2.lat.m}", flush=True)
