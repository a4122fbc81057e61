"""Round 6: the rectangular operator of the evaluation by itself -- K(x*, x) @ V with 101 columns (mean + 100 variance
columns), N = 1e6, d = 8, n* = N / 4 -- three calls, for a kernel trace.  python tools/rect_mvm_r6.py [n] [d] [cols]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers                                        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 101
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).to(dev)
xs = torch.randn(n // 4, d, generator=g).to(dev)
V = torch.randn(n, cols, generator=g).to(dev)
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).to(dev)
with torch.no_grad():
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K_star = model.kernel(xs, x)
        out = K_star.matmul(V)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out2 = K_star.matmul(V)                     # the same operator again: the lattice is in the cache
        torch.cuda.synchronize()
        print(f"rep {rep}: K(x*, x) @ V [{n // 4} x {n}] @ [{n} x {cols}]: first call {1e3 * (t1 - t0):.2f} ms, again {1e3 * (time.perf_counter() - t1):.2f} ms", flush=True)
    lat = list(plx.lattice_cache()._entries.values())[-1][0]
    print("m =", lat.m, "build ms", lat.build_times_ms())
