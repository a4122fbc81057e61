#!/usr/bin/env python3
"""Round 6: what keeping the point order across a re-scaled rebuild (Lattice.build(..., reuse_order=True)) saves the build
and costs the MVMs.  N = 1e6, d = 8: the order is computed at lengthscale l0 = 0.6931, the lattice is then rebuilt at
l0 * ratio with the order kept, against a cold build at l0 * ratio.  Build wall time (synchronised), one 12-column MVM in
lattice row order (a CG iteration's), one single-column MVM in caller order; outputs compared.  Interleaved, minimum over rounds.

    python tools/ab_reuse_order_r6.py [rounds]
"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
import bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, d, l0 = 1_000_000, 8, 0.6931
x, v = bench.synth(n, d, 12)
xc = x.cuda()
v12 = v.cuda().contiguous()
v1 = v12[:, :1].contiguous()


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def timed(fn, reps=10):
    fn()
    t0 = sync()
    for _ in range(reps):
        fn()
    return (sync() - t0) / reps * 1e6


for ratio in (1.0, 1.05, 1.1, 1.25, 1.5, 2.0, 0.8):
    ref0 = (xc / l0).contiguous()
    ref1 = (xc / (l0 * ratio)).contiguous()
    cold, warm = plx.Lattice(), plx.Lattice()
    res = {"cold": {}, "warm": {}}
    outs = {}
    for _ in range(rounds):
        for name, lat in (("cold", cold), ("warm", warm)):
            if name == "warm":
                lat.build(ref0, bench.RBF1)
            t0 = sync()
            lat.build(ref1, bench.RBF1, reuse_order=(name == "warm"))
            t1 = sync()
            r = res[name]
            r["build_ms"] = min(r.get("build_ms", 1e9), (t1 - t0) * 1e3)
            r["order_age"] = lat.order_age
            r["m"] = lat.m
            lat.prepare(1)
            outs[name] = lat.apply(v1).clone()
            r["mvm_vd1_us"] = min(r.get("mvm_vd1_us", 1e9), timed(lambda: lat.apply(v1)))
            lat.set_lattice_row_order(True)
            lat.apply(v12)
            r["mvm_vd12_lattice_rows_us"] = min(r.get("mvm_vd12_lattice_rows_us", 1e9), timed(lambda: lat.apply(v12)))
            lat.set_lattice_row_order(False)
    rel = float((outs["warm"] - outs["cold"]).norm() / outs["cold"].norm())
    print(json.dumps({"ratio": ratio, "cold": {k: round(val, 3) if isinstance(val, float) else val for k, val in res["cold"].items()},
                      "warm": {k: round(val, 3) if isinstance(val, float) else val for k, val in res["warm"].items()},
                      "out_rel_diff": rel}), flush=True)
    cold.close(); warm.close()
