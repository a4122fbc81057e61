#!/usr/bin/env python3
"""A/B the apply-kernel variants in ONE process (interleaved rounds), N=1e6 d=8.

Usage: python tools/ab_apply.py [--ell 1.0 0.25] [--vd 1] [--rounds 5]
Prints per-stage device time (torch events around 20 back-to-back launches) for
every value of every tunable, and checks that variants agree.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx  # noqa: E402
from simplex_gp_amd import _native as nv  # noqa: E402

RBF1 = np.array([0.34608543, 1.0, 0.34608543], np.float32)


def tune(key, val):
    nv.check(nv.lib().plx_tune(key.encode(), val), "plx_tune")


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ell", type=float, nargs="+", default=[1.0, 0.25])
    ap.add_argument("--vd", type=int, default=1)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.n, args.d, generator=g)
    v = torch.randn(args.n, args.vd, generator=g).cuda()
    variants = {"splat": [("splat_impl", 0), ("splat_impl", 1)],
                "blur": [("blur_vpt", 1), ("blur_vpt", 2), ("blur_vpt", 4)],
                "slice": [("slice_impl", 0), ("slice_impl", 1)]}
    for ell in args.ell:
        lat = plx.Lattice().build((x / ell).contiguous().cuda(), RBF1)
        m = lat.m
        vals = lat.new_values(args.vd)
        scratch = lat.new_values(args.vd)
        out = torch.empty_like(v)
        print(f"--- ell={ell} m={m} vd={args.vd}")
        stage_fn = {"splat": lambda: lat.splat(v, vals), "blur": lambda: lat.blur(vals, scratch),
                    "slice": lambda: lat.slice(vals, out)}
        for stage, vs in variants.items():
            res = {kv: [] for kv in vs}
            outs = {}
            for _ in range(args.rounds):
                for kv in vs:
                    tune(*kv)
                    res[kv].append(timeit(stage_fn[stage]))
            for kv in vs:
                tune(*kv)
                lat.splat(v, vals)
                if stage == "splat":
                    outs[kv] = vals.clone()
                elif stage == "blur":
                    outs[kv] = lat.blur(vals, scratch).clone()
                else:
                    outs[kv] = lat.slice(vals, out).clone()
            base = outs[vs[0]]
            for kv in vs:
                t = res[kv]
                err = (outs[kv] - base).norm().item() / base.norm().item()
                print(f"{stage:6s} {kv[0]}={kv[1]}: median {np.median(t):8.2f} us  min {min(t):8.2f} us   rel diff vs first {err:.2e}")
        # restore shipped defaults
        tune("splat_impl", 1); tune("blur_vpt", 4); tune("slice_impl", 1)
        full = timeit(lambda: lat.apply(v, out))
        print(f"apply (defaults): {full:.2f} us  -> {1e6 / full:.0f} MVM/s")
        lat.close()


if __name__ == "__main__":
    main()
