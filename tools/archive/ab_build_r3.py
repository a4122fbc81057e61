#!/usr/bin/env python3
"""Wall time of plx_build + plx_prepare(vd) (ms, best of several, synchronised at both ends) for plx_tune variants.
    python tools/ab_build_r3.py --n 1000000 --ell 1.0 --variants "order_compact=0" "order_compact=1" """
import argparse, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--vd", type=int, default=1)
ap.add_argument("--ell", type=float, default=1.0)
ap.add_argument("--variants", nargs="*", default=[""])
args = ap.parse_args()
x, v = bench.synth(args.n, args.d, args.vd)
ref = (x / args.ell).contiguous().cuda()
DEFAULTS = {"order_compact": 1, "block_e": 0, "vertex_order": 1, "sort_points": 1, "order_zcurve": 1, "insert_dedupe": 2}
lat = plx.Lattice()
for var in args.variants:
    for k, val in DEFAULTS.items():
        nv.check(nv.lib().plx_tune(k.encode(), val), "plx_tune")
    for kv in var.split():
        k, val = kv.split("=")
        nv.check(nv.lib().plx_tune(k.encode(), int(val)), "plx_tune")
    best_b, best_p = 1e9, 1e9
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lat.build(ref, bench.RBF1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        lat.prepare(args.vd)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if rep >= 2:
            best_b, best_p = min(best_b, (t1 - t0) * 1e3), min(best_p, (t2 - t1) * 1e3)
    lat.set_timing(True); lat.build(ref, bench.RBF1); st = lat.build_times_ms(); lat.set_timing(False)
    print(json.dumps({"variant": var or "default", "m": lat.m, "build_ms": round(best_b, 3), "prepare_ms": round(best_p, 3),
                      "stages_ms": {k: round(t, 3) for k, t in st.items()}}), flush=True)
