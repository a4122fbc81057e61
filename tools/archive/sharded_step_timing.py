#!/usr/bin/env python3
"""What one rank of a W-rank weak-scaling job (1e6 points per rank, bench.py --gpus W) executes, timed on ONE GPU:
all W local builds are run here one after the other to obtain the key sets, then rank 0's build_local + build_merge
and its splat / blur / slice over the merged lattice are timed.  Collectives are not included."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_local, d, ell = (int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000), 8, 1.0
from simplex_gp_amd import _native as nv
for kv in sys.argv[3:]:                      # plx_tune pairs: vertex_order=0 ...
    k, val = kv.split("=")
    nv.check(nv.lib().plx_tune(k.encode(), int(val)), "plx_tune")
g = torch.Generator().manual_seed(1234)
x = torch.randn(n_local * W, d, generator=g) / ell
v = torch.randn(n_local, 1, generator=g).cuda()
lat = plx.Lattice()
keys, counts = [], []
for r in range(W):
    k = lat.build_local(x[r * n_local:(r + 1) * n_local].contiguous().cuda(), RBF1)
    keys.append(k.clone()); counts.append(k.shape[0])
all_keys = torch.cat(keys, 0)
x0 = x[:n_local].contiguous().cuda()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
best = {}
for rep in range(4):
    t0 = sync(); lat.build_local(x0, RBF1); t1 = sync(); lat.build_merge(all_keys, counts, 0, total_points=n_local * W); t2 = sync()
    best["build_local_ms"] = min(best.get("build_local_ms", 9e9), (t1 - t0) * 1e3)
    best["build_merge_ms"] = min(best.get("build_merge_ms", 9e9), (t2 - t1) * 1e3)
lat.set_timing(True)
lat.build_local(x0, RBF1); lat.build_merge(all_keys, counts, 0, total_points=n_local * W); torch.cuda.synchronize()
best["merge_stage_ms"] = {k: round(v, 3) for k, v in lat.build_times_ms().items()}
lat.set_timing(False)
vals, scr = lat.new_values(1), lat.new_values(1)
out = torch.empty(n_local, 1, device="cuda")
best["splat_us"] = min(timeit(lambda: lat.splat(v, vals), iters=10) for _ in range(3))
best["blur_us"] = min(timeit(lambda: lat.blur(vals, scr, vd=1), iters=10) for _ in range(3))
res = lat.blur(vals, scr, vd=1)
best["slice_us"] = min(timeit(lambda: lat.slice(res, vd=1, out=out), iters=10) for _ in range(3))
best.update({"ranks": W, "m_union": lat.m, "local_vertex_counts": counts[:2], "key_exchange_MB": round(all_keys.numel() * 4 / 1e6, 1),
             "allreduce_MB": round(lat.m * 4 / 1e6, 2)})
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in best.items()}))
