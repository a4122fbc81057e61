"""Round 6: the training step's solve at the configs[4] stand-in (12 columns, rank-100 preconditioner, cg_tol 1) under cProfile:
where the HOST spends the 1.9 ms.  python tools/solve_hostprofile_r6.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers                                        # noqa: E402

n, d = 10623, 18
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).to(dev)
rhs = torch.randn(n, 11, generator=g).to(dev)
model = solvers.LatticeGP(plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d), min_noise=0.1).to(dev)
solvers.cap_host_threads()
with torch.no_grad():
    K = model.kernel(x, x)
    pre = model.preconditioner(x, 100, K=K)
    for _ in range(3):
        sol, info = model.khat_solve(x, rhs, K=K, max_iter=500, tol=1.0, want_tridiag=True, precond=pre)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(20):
        sol, info = model.khat_solve(x, rhs, K=K, max_iter=500, tol=1.0, want_tridiag=True, precond=pre)
    torch.cuda.synchronize()
    pr.disable()
    print(f"solve: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms, iterations {info['iterations']}")
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    st.print_stats(18)
