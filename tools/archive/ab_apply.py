#!/usr/bin/env python3
"""Stage times of the apply kernels, N=1e6 d=8, for plx_tune variants, in ONE process.

Usage: python tools/ab_apply.py [--ell 1.0 0.25] [--vd 1] [--rounds 3]
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


def tune(key, val, lat=None):
    """Set the process default of a switch (what the next build starts from) and, with `lat`, the copy of that built
    lattice as well (its MVMs from the next call on)."""
    nv.check(nv.lib().plx_tune(key.encode(), val), "plx_tune")
    if lat is not None:
        lat.tune(key, val)


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
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--sort", type=int, nargs="+", default=[0, 1])
    args = ap.parse_args()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.n, args.d, generator=g)
    v = torch.randn(args.n, args.vd, generator=g).cuda()
    for ell in args.ell:
        ref = (x / ell).contiguous().cuda()
        base = None
        for sort in args.sort:
            tune("sort_points", sort)
            lat = plx.Lattice()
            lat.set_timing(True)
            lat.build(ref, RBF1)
            bt = lat.build_times_ms()
            lat.set_timing(False)
            vals, scratch, out = lat.new_values(args.vd), lat.new_values(args.vd), torch.empty_like(v)
            lat.splat(v, vals)
            ts = min(timeit(lambda: lat.splat(v, vals)) for _ in range(args.rounds))
            tb = min(timeit(lambda: lat.blur(vals, scratch, vd=args.vd)) for _ in range(args.rounds))
            tl = min(timeit(lambda: lat.slice(vals, out, vd=args.vd)) for _ in range(args.rounds))
            ta = min(timeit(lambda: lat.apply(v, out)) for _ in range(args.rounds))
            res = lat.apply(v).clone()
            base = res if base is None else base
            err = (res - base).norm().item() / base.norm().item()
            print(f"ell={ell} vd={args.vd} m={lat.m} sort={sort}: splat {ts:7.2f} blur {tb:7.2f} slice {tl:7.2f} "
                  f"apply {ta:7.2f} us ({1e6 / ta:.0f} MVM/s) diff {err:.1e} | build "
                  + " ".join(f"{k}={t:.3f}" for k, t in bt.items()) + f" total={sum(bt.values()):.3f} ms", flush=True)
            lat.close()
        tune("sort_points", 1)


if __name__ == "__main__":
    main()
