#!/usr/bin/env python3
"""Randomised check of plx_filter_onehot: kernel rows on the frontier of their non-zero vertex rows against (1) the dense
splat_onehot + blur + slice stages (equal bits for multi-column rows: same operations in the same order), (2) the CPU
oracle's filter of the one-hot right-hand side.  Dimensions 1..20, tap orders 1..3, 1..16 columns of which 1..vd are
one-hot, both row orders, lengthscales from lattices of a few hundred vertices (the frontier covers everything) to ones
with more vertices than points; repeated calls on one lattice (the position map is cleared per call).
    python tests/checks/fuzz_onehot.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TAPS = {1: [0.34608543, 1.0, 0.34608543], 2: [0.08263808, 0.53616077, 1, 0.53616077, 0.08263808],
        3: [0.01831428, 0.16900772, 0.64117509, 1, 0.64117509, 0.16900772, 0.01831428]}
oracle.set_exact_mode(False)          # the duplicate-free lattice (what the product builds by default)
worst = 0.0
frontier = torch.zeros(1, dtype=torch.int32, device="cuda")
covered = 0
for c in range(cases):
    d = int(rng.integers(1, 21))
    n = int(rng.integers(200, 40000))
    order = int(rng.integers(1, 4))
    ell = float(rng.choice([0.1, 0.25, 0.5, 1.0, 2.0, 4.0]))
    taps = np.array(TAPS[order], np.float32)
    ref = (rng.standard_normal((n, d)) / ell).astype(np.float32)
    lat = plx.Lattice().build(torch.from_numpy(ref).cuda(), taps)
    perm = lat.export(nv.ARRAY_POINT_PERM).astype(np.int64)
    for rep in range(3):
        vd = int(rng.choice([1, 4, 8, 12, 16]))
        nb = int(rng.integers(1, vd + 1))
        lattice_rows = bool(rng.integers(0, 2))
        lat.set_lattice_row_order(lattice_rows)
        pts = rng.integers(0, n, nb).astype(np.int32)
        if nb > 1 and rng.integers(0, 2):
            pts[1] = (pts[0] + 1) % n
        pd = torch.from_numpy(pts).cuda()
        vals, scratch = lat.new_values(vd), lat.new_values(vd)
        dense = torch.full((n, vd), 3.0, device="cuda")
        got = torch.full((n, vd), 5.0, device="cuda")
        lat.filter_onehot(pd, nb, vals, scratch, dense, vd=vd, sparse=False)
        lat.filter_onehot(pd, nb, vals, scratch, got, vd=vd, sparse=True, frontier=frontier)
        f = int(frontier.item())
        assert 1 <= f <= lat.m, (c, f, lat.m)
        covered += f == lat.m
        if vd > 1:
            assert torch.equal(got, dense), (c, rep, n, d, order, vd, nb, lattice_rows, float((got - dense).abs().max()))
        src = np.zeros((n, vd), np.float32)
        rows = pts.astype(np.int64) if lattice_rows else perm[pts]
        # the oracle works in the caller's row order: lattice-order rows are caller rows perm[.]
        src_caller = np.zeros((n, vd), np.float32)
        src_caller[perm[pts], np.arange(nb)] = 1.0
        want_caller = oracle.filter(src_caller, ref, taps)
        want = want_caller[perm] if lattice_rows else want_caller
        nrm = np.linalg.norm(want.astype(np.float64)) + 1e-30
        err = np.linalg.norm(got.cpu().numpy().astype(np.float64) - want) / nrm
        worst = max(worst, err)
        assert err < 2e-5, (c, rep, n, d, order, vd, nb, lattice_rows, err)
    lat.set_lattice_row_order(False)
    lat.close()
    if c % 20 == 19:
        print(f"{c + 1} lattices: worst rel-L2 vs oracle {worst:.2e}; calls whose frontier covered the lattice: {covered}", flush=True)
print(f"fuzz_onehot: {cases} lattices x 3 calls, frontier == dense stages bit for bit (multi-column), worst rel-L2 vs oracle {worst:.2e}, "
      f"frontier covered the whole lattice in {covered} calls")
