#!/usr/bin/env python3
"""BASELINE.json configs[2] under rocprofv3: one lattice build + `--iters` CG iterations at N=1e6, d=8, vd=11.

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 <repo>/tools/prof_cg.py
    python3 tools/prof_mvm.py --stats <out>
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ell", type=float, default=0.6931)
    ap.add_argument("--tune", nargs="*", default=[])
    ap.add_argument("--no-fused-steps", action="store_true", help="the CG iteration with its two stand-alone reductions (what solvers does by itself above FUSED_CG_MAX_ROWS rows)")
    ap.add_argument("--fused-steps", action="store_true", help="the reductions inside the update / direction kernels at any size")
    args = ap.parse_args()
    import torch
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers, _native as nv
    if args.no_fused_steps:
        solvers.FUSED_CG_STEPS = False
    if args.fused_steps:
        solvers.FUSED_CG_STEPS = True
    for kv in args.tune:
        k, v = kv.split("=")
        nv.check(nv.lib().plx_tune(k.encode(), int(v)), "plx_tune")
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.n, args.d, generator=g).cuda()
    y = torch.randn(args.n, generator=g).cuda()
    Z = (torch.randint(0, 2, (args.n, 10), generator=g).float() * 2 - 1).cuda()
    rhs = torch.cat([y[:, None], Z], 1)
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=args.d)).cuda()
    with torch.no_grad():
        for trial in range(3):
            model.kernel.lengthscale = args.ell * (1 + 1e-5 * trial)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sol, info = model.khat_solve(x, rhs, max_iter=args.iters, tol=0.0)
            torch.cuda.synchronize()
            print(f"trial {trial}: {args.iters} CG iterations incl. build {1e3 * (time.perf_counter() - t0):.2f} ms, "
                  f"residual max {float(info['residual'].max()):.3e}")


if __name__ == "__main__":
    main()
