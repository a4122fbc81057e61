#!/usr/bin/env python3
"""Round 4: fine / medium regime build and splat A/B (N=1e6, d=8): neighbour lookups with the slot-occupancy bitmap
(nbr_bitmap) and the first-touch splat (splat_first), interleaved in one process.
    python tools/ab_fine_r4.py [ell ...]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench

ells = [float(a) for a in sys.argv[1:]] or [0.25, 0.6931]
n, d = 1_000_000, 8
x, v = bench.synth(n, d, 1)
vc = v.cuda()
out = torch.empty_like(vc)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for ell in ells:
    ref = (x / ell).contiguous().cuda()
    res = {}
    outs = {}
    for rnd in range(3):
        for name, tunes in (("base", {"nbr_bitmap": 0, "splat_first": 0}), ("bitmap", {"nbr_bitmap": 2, "splat_first": 0}),
                            ("bitmap+first", {"nbr_bitmap": 2, "splat_first": 1})):
            for k, val in tunes.items():
                nv.check(nv.lib().plx_tune(k.encode(), val), "tune")
            lat = res.setdefault(name, {}).get("lat") or plx.Lattice()
            res[name]["lat"] = lat
            lat.set_timing(True)
            t0 = sync()
            lat.build(ref, bench.RBF1)
            t1 = sync()
            bt = lat.build_times_ms()
            lat.set_timing(False)
            t2 = sync()
            lat.prepare(1)
            t3 = sync()
            for _ in range(3):
                lat.apply(vc, out)
            t4 = sync()
            for _ in range(20):
                lat.apply(vc, out)
            t5 = sync()
            lat.set_timing(True)
            st = {"splat": [], "blur": [], "slice": []}
            for _ in range(10):
                lat.apply(vc, out)
                t = lat.apply_times_ms()
                for kk in st:
                    st[kk].append(t[kk])
            lat.set_timing(False)
            r = res[name]
            r["build_ms"] = min(r.get("build_ms", 1e9), (t1 - t0) * 1e3)
            r["nbr_ms"] = min(r.get("nbr_ms", 1e9), bt["neighbours"])
            r["stages"] = {kk: round(float(bt[kk]), 3) for kk in bt}
            r["tables_ms"] = min(r.get("tables_ms", 1e9), (t3 - t2) * 1e3)
            r["mvm_us"] = min(r.get("mvm_us", 1e9), (t5 - t4) / 20 * 1e6)
            r["splat_us"] = min(r.get("splat_us", 1e9), float(np.mean(st["splat"])) * 1e3)
            r["kernels"] = lat.stage_kernels()["splat"]
            r["m"] = lat.m
            outs[name] = out.clone()
    for name, r in res.items():
        r.pop("lat").close()
        print(json.dumps({"ell": ell, "variant": name, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}}), flush=True)
    base = outs["base"]
    for name, o in outs.items():
        print(json.dumps({"ell": ell, "variant": name, "rel_diff_vs_base": float((o - base).norm() / base.norm()),
                          "bit_identical": bool(torch.equal(o, base))}), flush=True)
    del ref
for k, val in (("nbr_bitmap", 1), ("splat_first", 1)):
    nv.lib().plx_tune(k.encode(), val)
