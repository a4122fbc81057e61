#!/usr/bin/env python3
"""The bench's timed loop (one build + K MVMs, no synchronisation inside) for plx_tune variants: us per step.
    python tools/ab_loop_r3.py --steps 20 --variants "order_compact=0" "order_compact=1" "readback_spin=0" """
import argparse, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--ell", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--variants", nargs="*", default=[""])
args = ap.parse_args()
x, v = bench.synth(args.n, 8, 1)
ref = (x / args.ell).contiguous().cuda(); v = v.cuda(); out = torch.empty_like(v)
DEFAULTS = {"order_compact": 1, "readback_spin": 1, "block_e": 0}
lat = plx.Lattice()
for var in args.variants:
    for k, val in DEFAULTS.items():
        nv.check(nv.lib().plx_tune(k.encode(), val), "plx_tune")
    for kv in var.split():
        k, val = kv.split("=")
        nv.check(nv.lib().plx_tune(k.encode(), int(val)), "plx_tune")
    def loop():
        for i in range(args.steps):
            if i == 0:
                lat.build(ref, bench.RBF1)
            lat.apply(v, out)
    for _ in range(3):
        loop()
    best = 1e9
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loop(); loop(); loop()                      # the second and third builds queue behind running MVMs, as in a CG loop
        torch.cuda.synchronize(); t1 = time.perf_counter()
        best = min(best, (t1 - t0) / (3 * args.steps) * 1e6)
    print(json.dumps({"variant": var or "default", "us_per_step": round(best, 1)}), flush=True)
