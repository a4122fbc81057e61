#!/usr/bin/env python3
"""Host time per Lattice.apply() call (enqueue only) vs GPU time, small and large lattices."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1
for (n, d) in [(200, 1), (100000, 4), (1000000, 8)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, d, generator=g).cuda(); v = torch.randn(n, 1, generator=g).cuda()
    lat = plx.Lattice().build(x, RBF1); out = torch.empty_like(v)
    for _ in range(20): lat.apply(v, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): lat.apply(v, out)
    host = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    gpu = timeit(lambda: lat.apply(v, out), iters=200)
    print(f"n={n} d={d}: host enqueue {host:6.1f} us/call, end-to-end {gpu:6.1f} us/call", flush=True)
    lat.close()
