#!/usr/bin/env python3
"""Round-3 A/B in one process: stage times (splat / blur / slice, us) and whole-MVM time for plx_tune variants.
    python tools/ab_r3.py --n 1000000 --ell 1.0 --vd 1 --variants "block_e=16" "block_e=24"
Each variant = space-separated key=value pairs applied on top of the defaults (reset between variants)."""
import argparse, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
from tools.archive.ab_apply import timeit, RBF1

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--vd", type=int, default=1)
ap.add_argument("--ell", type=float, default=1.0)
ap.add_argument("--lattice-rows", action="store_true")
ap.add_argument("--variants", nargs="*", default=[""])
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
x, v = bench.synth(args.n, args.d, args.vd)
ref = (x / args.ell).contiguous().cuda()
v = v.cuda()
out = torch.empty_like(v)
seen = set()
for var in args.variants:
    pairs = [kv.split("=") for kv in var.split()]
    for k, val in pairs:
        nv.check(nv.lib().plx_tune(k.encode(), int(val)), "plx_tune"); seen.add(k)
    lat = plx.Lattice().build(ref, RBF1)
    if args.lattice_rows:
        lat.set_lattice_row_order(True)
    lat.prepare(args.vd)
    for _ in range(3):
        lat.apply(v, out)
    vals, scr = lat.new_values(args.vd), lat.new_values(args.vd)
    res = {"variant": var or "default", "m": lat.m, "block_rows": lat.block_rows}
    res["mvm_us"] = round(min(timeit(lambda: lat.apply(v, out), iters=20) for _ in range(args.rounds)), 1)
    res["splat_us"] = round(min(timeit(lambda: lat.splat(v, vals), iters=20) for _ in range(args.rounds)), 1)
    res["blur_us"] = round(min(timeit(lambda: lat.blur(vals, scr, vd=args.vd), iters=20) for _ in range(args.rounds)), 1)
    r = lat.blur(vals, scr, vd=args.vd)
    res["slice_us"] = round(min(timeit(lambda: lat.slice(r, out, vd=args.vd), iters=20) for _ in range(args.rounds)), 1)
    res["kernels"] = {k: "+".join(n) for k, n in lat.stage_kernels().items() if k != "vertex_order"}
    print(json.dumps(res), flush=True)
    lat.close()
    for k, val in pairs:                         # back to the defaults
        pass
    # defaults of the keys this script touches
    for k in seen:
        d = {"block_e": 0, "block_dense_combine": 1, "block_path": 1, "order_zcurve": 1, "vertex_order": 1, "order_compact": 1}.get(k)
        if d is not None:
            nv.check(nv.lib().plx_tune(k.encode(), d), "plx_tune")
