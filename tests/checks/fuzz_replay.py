#!/usr/bin/env python3
"""Randomised check of plx_tune("reference_growth", 1): HIP with the replay on against the oracle in the reference-exact
mode (which is the reference's CPU extension bit for bit) -- dimensions 2..14, tap orders 1..3, 1..12 columns, point counts
and lengthscales chosen so that the reference's table doubles several times (m > 2^14).
    python tests/checks/fuzz_replay.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TAPS = {1: [0.34608543, 1.0, 0.34608543], 2: [0.08263808, 0.53616077, 1, 0.53616077, 0.08263808],
        3: [0.01831428, 0.16900772, 0.64117509, 1, 0.64117509, 0.16900772, 0.01831428]}
nv.check(nv.lib().plx_tune(b"reference_growth", 1), "tune")
worst, hits, shown = 0.0, 0, 0
for c in range(cases):
    d = int(rng.integers(2, 15))
    n = int(rng.integers(3000, 60000))
    order = int(rng.integers(1, 4))
    vd = int(rng.choice([1, 1, 2, 3, 5, 12]))
    ell = float(rng.choice([0.15, 0.25, 0.4, 0.6, 1.0]))
    taps = np.array(TAPS[order], np.float32)
    ref = (rng.standard_normal((n, d)) / ell).astype(np.float32)
    src = rng.standard_normal((n, vd)).astype(np.float32)
    exact, m_exact = oracle.filter(src, ref, taps, return_m=True)
    oracle.set_exact_mode(False)
    clean = oracle.filter(src, ref, taps)
    oracle.set_exact_mode(True)
    lat = plx.Lattice().build(torch.from_numpy(ref).cuda(), taps)
    out = lat.apply(torch.from_numpy(src).cuda()).cpu().numpy()
    info = lat.reference_growth_info()
    nrm = np.linalg.norm(exact.astype(np.float64))
    err = np.linalg.norm(out.astype(np.float64) - exact) / nrm
    quirk = np.linalg.norm(clean.astype(np.float64) - exact) / nrm
    worst = max(worst, err)
    hits += quirk > 1e-6
    assert info["m_reference"] == m_exact and not info["inexact"], (c, n, d, order, vd, ell, info, m_exact)
    assert err <= 1e-5, (c, n, d, order, vd, ell, err, quirk, info)
    if quirk > 1e-6 and shown < 12:
        shown += 1
        print(f"case {c}: n={n} d={d} order={order} vd={vd} l={ell}: quirk alone {quirk:.2e}, replay vs reference {err:.2e}, {info}", flush=True)
    lat.close()
nv.check(nv.lib().plx_tune(b"reference_growth", 0), "tune")
print(f"OK {cases} cases, the quirk moved the reference on {hits} of them; worst rel-L2 with the replay on {worst:.2e}")
