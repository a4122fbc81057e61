#!/usr/bin/env python3
"""Round 6: the two backward-pass questions of the round-5 verdict, variants interleaved in ONE process (rule 24),
minimum over rounds.  N = 1e6, d = 8, L = 11 (198 columns), lattices of the training step (l = 0.6931 -> m = 1.73e6 and the
second step's l ~ 0.78 -> m ~ 1.0e6).

  (1) slice + contraction: plx_tune("contract_v", 0 / 1) -- the run-time kernel against the one with the corner count
      compiled in (all rows in flight, all-lane contraction).  Stage times from the lattice's own
      events (splat / blur / slice of the fused backward); gradients compared.
  (2) the 198-column blur tile-outer / axis-inner: the SAME d+1 passes run tile by tile over column tiles of T columns
      laid out [tiles][m][T] (each tile's ping-pong pair fits the 256 MiB Infinity Cache: 2 x m x 4T bytes) through the
      kernels that serve narrow rows (T = 8, 12, 16: the two-axes-per-launch kernels of the CG iteration; 32 / 64: the
      general row kernel), against one blur of [m][200].  This is what a [T][m][16] value layout could buy the blur
      BEFORE the splat / slice kernels pay for producing / consuming it.

    python tools/ab_backward_r6.py [n] [rounds]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d, L = 8, 11
W = 2 * L * (1 + d)
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
taps = plx.DiscretizedKernelFN(plx.rbf, 1).get_deriv_coeffs()


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def timed(fn, reps):
    fn()
    t0 = sync()
    for _ in range(reps):
        fn()
    return (sync() - t0) / reps * 1e6


for ell in (0.6931, 0.78):
    ref = (x / ell).contiguous().cuda()
    src = torch.randn(n, L, generator=g).cuda()
    go = torch.randn(n, L, generator=g).cuda()

    # ---- (1) slice + contraction ------------------------------------------------------------------
    lats, grads = {}, {}
    for v in (0, 1):
        nv.check(nv.lib().plx_tune(b"contract_v", v), "plx_tune")
        lats[v] = plx.Lattice().build(ref, taps)
        lats[v].set_lattice_row_order(True)
        lats[v].set_timing(True)
        grads[v] = lats[v].apply_backward(go, src, ref)
    nv.check(nv.lib().plx_tune(b"contract_v", 1), "plx_tune")
    best = {0: None, 1: None}
    wall = {0: 1e9, 1: 1e9}
    for _ in range(rounds):
        for v in (0, 1):
            t0 = sync()
            lats[v].apply_backward(go, src, ref)
            t1 = sync()
            st = lats[v].apply_times_ms()
            wall[v] = min(wall[v], (t1 - t0) * 1e3)
            if best[v] is None or st["slice"] < best[v]["slice"]:
                best[v] = st
    gx0, gs0 = grads[0]
    gx1, gs1 = grads[1]
    rel = float((gx1 - gx0).norm() / gx0.norm())
    print(json.dumps({"what": "slice_contract", "ell": ell, "m": lats[0].m,
                      "runtime_form": {"stages_ms": best[0], "backward_ms": round(wall[0], 3)},
                      "compiled_form": {"stages_ms": best[1], "backward_ms": round(wall[1], 3)},
                      "grad_x_rel_diff": rel, "grad_src_equal": bool(torch.equal(gs0, gs1))}), flush=True)
    lat = lats[1]
    lats[0].close()
    lat.set_timing(False)
    m = lat.m

    # ---- (2) the wide blur, tile-outer / axis-inner ----------------------------------------------------
    vdp = lat.values_stride(W)
    wide_a = torch.randn(m, vdp, device="cuda")
    wide_b = torch.empty_like(wide_a)
    variants = {"wide_[m][200]": None}
    tiles = {}
    for T in (8, 12, 16, 32, 64):
        K = -(-W // T)
        tiles[T] = (torch.randn(K, m, T, device="cuda"), torch.empty(m, T, device="cuda"), K)
        variants[f"tiles_{K}x[m][{T}]"] = T
    # the tiled form computes the same numbers: check one tile against the same columns of the wide blur
    wide_a[:, :16].copy_(tiles[16][0][0])
    ref_cols = lat.blur(wide_a.clone(), wide_b, vd=W)[:, :16].clone()
    got = lat.blur(tiles[16][0][0].clone(), tiles[16][1], vd=16)
    same = bool(torch.equal(ref_cols, got))

    def run(T):
        if T is None:
            lat.blur(wide_a, wide_b, vd=W)
        else:
            buf, scr, K = tiles[T]
            for t in range(K):
                lat.blur(buf[t], scr, vd=T)

    res = {k: 1e18 for k in variants}
    for _ in range(rounds):
        for k, T in variants.items():
            res[k] = min(res[k], timed(lambda: run(T), 2))
    print(json.dumps({"what": "blur_198_columns", "ell": ell, "m": m, "tile16_bit_equal_to_wide": same,
                      "us": {k: round(v, 1) for k, v in res.items()},
                      "pingpong_pair_MB": {str(T): round(2 * m * 4 * T / 2**20, 1) for T in (8, 12, 16, 32, 64)}}), flush=True)
    lat.close()
    del wide_a, wide_b, tiles
    torch.cuda.empty_cache()
