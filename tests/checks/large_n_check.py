#!/usr/bin/env python3
"""One K.v at N = 1e7 (d = 8, l = 1): build / MVM time, memory, and rel-L2 against the CPU oracle."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from oracle import oracle
from tools.archive.ab_apply import timeit, RBF1
n, d = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); v = torch.randn(n, 1, generator=g)
xc, vc = x.cuda(), v.cuda()
lat = plx.Lattice()
t0 = time.perf_counter(); lat.build(xc, RBF1); torch.cuda.synchronize(); t1 = time.perf_counter()
out = lat.apply(vc)
tb = min(timeit(lambda: lat.build(xc, RBF1), iters=2) for _ in range(2))
ta = min(timeit(lambda: lat.apply(vc, out), iters=5) for _ in range(2))
res = {"n": n, "m": lat.m, "build_ms": round(tb / 1e3, 2), "mvm_us": round(ta, 1), "lattice_GB": round(lat.device_bytes / 1e9, 2)}
print(json.dumps(res), flush=True)
oracle.set_exact_mode(False)
t2 = time.perf_counter(); want = oracle.filter(v.numpy(), x.numpy(), RBF1); t3 = time.perf_counter()
got = out.cpu().numpy()
res.update({"oracle_s": round(t3 - t2, 1), "rel_l2_vs_oracle": float(np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want))})
print(json.dumps(res))
