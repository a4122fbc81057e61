#!/usr/bin/env python3
"""Round 4: what ONE rank executes in the column-sharded / grid modes of distributed.py, rehearsed on one GPU.
  A  config 3 (N=1e6, d=8, l=0.6931): 50 CG iterations on t of the 11 columns (t = the rank's share)
  B  config 4 (N=4e6, l=1): one MVM of a rank's column block on the whole operator (P = 1)
  C  config 4, P = 2: a rank's half of the rows (shard 0 of 2 of a replicated build: same union lattice), its column block:
     splat / blur / slice (the vertex all-reduce of values[m, vdp] sits between splat and blur and is not timed here)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers
import bench

def sync(): torch.cuda.synchronize(); return time.perf_counter()
def best_of(fn, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = sync(); fn(); b = min(b, sync() - t0)
    return b

which = sys.argv[1:] or ["A", "B", "C"]
if "A" in which:
    n, d = 1_000_000, 8
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).cuda()
    y = torch.randn(n, generator=g)
    Z = torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1
    rhs = torch.cat([y[:, None], Z], 1).cuda()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        K = model.kernel(x, x)
        for t in (1, 2, 3, 4, 6, 8, 11):
            B = rhs[:, :t].contiguous()
            model.khat_solve(x, B, K=K, max_iter=5, tol=0.0)
            dt = best_of(lambda: model.khat_solve(x, B, K=K, max_iter=50, tol=0.0))
            print(json.dumps({"leg": "A config3 CG", "columns": t, "ms_50_iterations": round(dt * 1e3, 2), "us_per_iteration": round(dt / 50 * 1e6, 1)}), flush=True)
        trial = [0]
        def with_build():
            trial[0] += 1
            model.kernel.lengthscale = 0.6931 * (1 + 1e-6 * trial[0])
            model.khat_solve(x, rhs[:, :2].contiguous(), max_iter=50, tol=0.0)
        print(json.dumps({"leg": "A config3 CG", "columns": 2, "ms_50_iterations_incl_build": round(best_of(with_build) * 1e3, 2)}), flush=True)
    plx.lattice_cache().clear()
    del x, rhs
if "B" in which or "C" in which:
    n, d = 4_000_000, 8
    x, v = bench.synth(n, d, 11)
    ref = x.cuda()
    if "B" in which:
        lat = plx.Lattice().build(ref, bench.RBF1)
        for vd in (1, 2, 3, 4, 6, 11):
            vb = v[:, :vd].contiguous().cuda(); out = torch.empty_like(vb)
            for _ in range(3): lat.apply(vb, out)
            dt = best_of(lambda: [lat.apply(vb, out) for _ in range(10)]) / 10
            print(json.dumps({"leg": "B config4 P=1", "columns": vd, "mvm_us": round(dt * 1e6, 1), "m": lat.m}), flush=True)
        lat.close()
    if "C" in which:
        for P in (2, 4):
            lat = plx.Lattice().build(ref, bench.RBF1, shard=(0, P))
            rows = lat.n_owned
            for vd in (1, 2, 3, 6, 11):
                vb = v[:rows, :vd].contiguous().cuda(); out = torch.empty_like(vb)
                vals, scr = lat.new_values(vd), lat.new_values(vd)
                for _ in range(2): lat.splat(vb, vals); r = lat.blur(vals, scr, vd=vd); lat.slice(r, out, vd=vd)
                ts = best_of(lambda: [lat.splat(vb, vals) for _ in range(10)]) / 10
                tb = best_of(lambda: [lat.blur(vals, scr, vd=vd) for _ in range(10)]) / 10
                r = lat.blur(vals, scr, vd=vd)
                tl = best_of(lambda: [lat.slice(r, out, vd=vd) for _ in range(10)]) / 10
                print(json.dumps({"leg": f"C config4 P={P}", "rows": rows, "columns": vd, "splat_us": round(ts * 1e6, 1), "blur_us": round(tb * 1e6, 1),
                                  "slice_us": round(tl * 1e6, 1), "allreduce_bytes": lat.m * lat.values_stride(vd) * 4, "m": lat.m}), flush=True)
            lat.close()
