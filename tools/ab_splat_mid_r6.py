"""Round 6: the splat of rows of 17..31 sixteen-byte chunks (65..124 columns: the evaluation's 101 = mean + 100 variance
columns) -- column tiles through splat_scan_kernel (plx_tune("splat_wide", 2): the 32-chunk gate of rounds 1-5) against the
row-parallel wide kernel with idle lanes (1, the default since), and the blur through the general kernel
(plx_tune("blur_multi", 2)) against the wide-row kernels -- four items per thread, or only the rows that change on a sparse
lattice -- (1); whole MVMs, interleaved.  python tools/ab_splat_mid_r6.py [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
CASES = [("N = 1.25e6, d = 8, RBF order 1, l = 0.6931 (the evaluation's stacked lattice at N = 1e6)", 1_250_000, 8, 0.6931, plx.RBFLattice(order=1)),
         ("N = 16,599, d = 18, Matern order 3, l = 1 (configs[4] stand-in, stacked)", 16_599, 18, 1.0, plx.MaternLattice(nu=1.5, order=3)),
         ("N = 61,000, d = 20, RBF order 1, l = 0.6931 (keggdirected shape, stacked)", 61_000, 20, 0.6931, plx.RBFLattice(order=1))]
MODES = [("column tiles + general blur", 2, 2), ("wide splat + general blur", 1, 2), ("wide splat + multi / active-row blur", 1, 1)]
for label, n, d, ell, kern in CASES:
    g = torch.Generator().manual_seed(1234)
    x = (torch.randn(n, d, generator=g) / ell).to(dev)
    taps = kern.dkernel_fn.get_coeffs().numpy()
    lat = plx.Lattice(dev).build(x, taps)
    print(f"{label}: m = {lat.m}", flush=True)
    for cols in (68, 101, 124):
        V = torch.randn(n, cols, generator=g).to(dev)
        out, best = {}, {}
        for r in range(rounds + 1):
            for name, sw, bm in MODES:
                lat.tune("splat_wide", sw)
                lat.tune("blur_multi", bm)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    o = lat.apply(V)
                torch.cuda.synchronize()
                if r:
                    best[name] = min(best.get(name, 1e9), (time.perf_counter() - t0) / 3 * 1e3)
                out[name] = o
        lat.tune("splat_wide", 1)
        lat.tune("blur_multi", 1)
        ref = out[MODES[0][0]]
        print(f"  cols {cols}: " + "; ".join(f"{name} {best[name]:.3f} ms (rel diff {float((out[name] - ref).abs().max() / ref.abs().max()):.1e})"
                                               for name, _, _ in MODES), flush=True)
    lat.close()
