#!/usr/bin/env python3
"""Round 4: neighbour lookups through a window of the sorted Morton codes (nbr_window) against the hash table alone, on
Morton-numbered lattices (N=1e6, d=8); build stage times and exported neighbour tables compared.
    python tools/ab_nbr_window_r4.py [ell ...]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench

ells = [float(a) for a in sys.argv[1:]] or [1.0, 0.6931, 0.5, 0.4]
x, v = bench.synth(1_000_000, 8, 1)
vc = v.cuda()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for ell in ells:
    ref = (x / ell).contiguous().cuda()
    res, tabs = {}, {}
    for rnd in range(3):
        for w in (0, 512, 2048, 8192):
            nv.check(nv.lib().plx_tune(b"nbr_window", w), "tune")
            lat = res.setdefault(w, {}).get("lat") or plx.Lattice()
            res[w]["lat"] = lat
            lat.set_timing(True)
            t0 = sync(); lat.build(ref, bench.RBF1); t1 = sync()
            bt = lat.build_times_ms()
            lat.set_timing(False)
            res[w]["build_ms"] = min(res[w].get("build_ms", 1e9), (t1 - t0) * 1e3)
            res[w]["nbr_ms"] = min(res[w].get("nbr_ms", 1e9), bt["neighbours"])
            res[w]["m"] = lat.m
            if rnd == 0:
                tabs[w] = lat.export(nv.ARRAY_NEIGHBORS)
                res[w]["out"] = lat.apply(vc).clone()
    for w, r in res.items():
        r.pop("lat").close()
        same = bool(np.array_equal(tabs[w], tabs[0])) and bool(torch.equal(r.pop("out"), res[0].get("out", r.get("out")))) if w else True
        print(json.dumps({"ell": ell, "nbr_window": w, "m": r["m"], "build_ms": round(r["build_ms"], 3), "neighbours_ms": round(r["nbr_ms"], 3),
                          "table_and_output_identical_to_hash_only": same}), flush=True)
    del ref
nv.lib().plx_tune(b"nbr_window", 512)
