#!/usr/bin/env python3
"""A bare MVM loop for rocprofv3: builds one lattice and applies it `--reps` times.

    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 <repo>/tools/prof_mvm.py --tune block_path=2
    python3 tools/prof_mvm.py --stats <out>      # prints the per-kernel table of that run
"""
import argparse
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def show(path):
    files = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    for f in files:
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            name = r["Name"].replace("void ", "").split("(")[0][:70]
            if float(r["Percentage"]) >= 0.3:
                print(f"{name:72s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  min {float(r['MinNs']) / 1e3:9.2f}  {float(r['Percentage']):5.1f} %")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", default=None)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--vd", type=int, default=1)
    ap.add_argument("--ell", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--lattice-rows", action="store_true")
    ap.add_argument("--builds", type=int, default=1, help="build the lattice this many times (timelines of a warm build)")
    ap.add_argument("--tune", nargs="*", default=[], help="key=value pairs for plx_tune")
    args = ap.parse_args()
    if args.stats:
        return show(args.stats)
    import numpy as np
    import torch
    import simplex_gp_amd as plx
    from simplex_gp_amd import _native as nv
    for kv in args.tune:
        k, v = kv.split("=")
        nv.check(nv.lib().plx_tune(k.encode(), int(v)), "plx_tune")
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.n, args.d, generator=g)
    v = torch.randn(args.n, args.vd, generator=g).cuda()
    ref = (x / args.ell).contiguous().cuda()
    lat = plx.Lattice()
    import time
    for _ in range(args.builds):
        t0 = time.perf_counter()
        lat.build(ref, np.array([0.34608543, 1.0, 0.34608543], np.float32))
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
    if args.lattice_rows:
        lat.set_lattice_row_order(True)
    out = torch.empty_like(v)
    for _ in range(args.reps):
        lat.apply(v, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        lat.apply(v, out)
    b.record()
    torch.cuda.synchronize()
    print(f"m={lat.m} build {build_ms:.3f} ms block_rows={lat.block_rows} apply {a.elapsed_time(b) / args.reps * 1e3:.1f} us  kernels {lat.stage_kernels()}")


if __name__ == "__main__":
    main()
