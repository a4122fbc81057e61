#!/usr/bin/env python3
"""Round 5: would LDS-staged column tiles pay in the backward pass?  Builds the lattice the backward
filter of a training step runs on (N = 1e6, d = 8, derivative taps of the RBF order-1 kernel, GPyTorch's default initial
lengthscale) and reports what a block formulation could share: distinct vertices per point block (block rows) against
corners, and the line traffic a column-tiled gather would move against the row-at-a-time gather of slice_contract_kernel.
    python tools/backward_block_rows.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv

n, d, L = 1_000_000, 8, 11
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
kern = plx.RBFLattice(order=1, ard_num_dims=d)
fwd, der = kern.dkernel_fn.get_coeffs().numpy(), kern.dkernel_fn.get_deriv_coeffs().numpy()
ref = (x / 0.6931).contiguous().cuda()
cols = 2 * L * (1 + d)
row_bytes = ((cols + 3) // 4 * 4) * 4
for name, taps in (("forward taps", fwd), ("derivative taps (the backward filter)", der)):
    for e in (16, 24):
        nv.check(nv.lib().plx_tune(b"block_e", e), "tune")
        nv.check(nv.lib().plx_tune(b"block_path", 2), "tune")
        lat = plx.Lattice().build(ref, taps)
        lat.prepare(1)
        rb, nnz, m = lat.block_rows, n * (d + 1), lat.m
        pts = 256 * e // (d + 1) // (48 if e == 24 else 16) * (48 if e == 24 else 16)
        rows_per_block = rb / (n / pts)
        # row-at-a-time: every corner gathers its vertex row (whole 128-byte lines: ceil(row_bytes / 128) lines)
        now = nnz * ((row_bytes + 127) // 128) * 128
        out = {"taps": name, "block_e": e, "points_per_block": pts, "m": m, "corners": nnz, "block_rows": rb,
               "block_rows_per_corner": round(rb / nnz, 3), "rows_per_block": round(rows_per_block, 1),
               "row_bytes": row_bytes, "gather_now_GB": round(now / 1e9, 2)}
        for tile_cols in (16, 32, 64):
            tile_b = tile_cols * 4
            ntiles = -(-cols // tile_cols)
            lds_kb = rows_per_block * tile_b / 1024
            # a tile of a row is tile_b contiguous bytes inside the row: it costs whole 128-byte lines
            lines = max(1, -(-tile_b // 128))
            traffic = rb * ntiles * lines * 128
            out[f"tile{tile_cols}"] = {"lds_KB_per_block": round(lds_kb, 1), "tiles": ntiles, "gather_GB": round(traffic / 1e9, 2)}
        print(json.dumps(out), flush=True)
        lat.close()
nv.lib().plx_tune(b"block_e", 0); nv.lib().plx_tune(b"block_path", 1)
