#!/usr/bin/env python3
"""Would another vertex numbering shorten the blur's neighbour distances?  Exports keys + neighbour table of a lattice and
compares the current numbering (first touch along the point order) with a Morton order of the vertices' own coordinates
in the blur-axis basis a_i = (k_d - k_i) / (d+1) (an axis step i < d changes a_i alone by +-1).
Prints, per numbering, the share of existing neighbours farther than a few thresholds (in vertex ids)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv

n, d = 1_000_000, 8
ell = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6931
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
lat = plx.Lattice().build((x / ell).contiguous().cuda(), np.array([0.34608543, 1.0, 0.34608543], np.float32))
keys = lat.export(nv.ARRAY_KEYS).astype(np.int64)            # [m, d]
nbr = lat.export(nv.ARRAY_NEIGHBORS)                         # [d+1, 2, m]
m = lat.m
kd = -keys.sum(1)
a = (kd[:, None] - keys) // (d + 1)
assert np.array_equal(a * (d + 1), kd[:, None] - keys)
a -= a.min(0)
bits = int(np.ceil(np.log2(a.max() + 1)))
code = np.zeros(m, np.uint64)
for b in range(bits - 1, -1, -1):
    for i in range(d):
        code = (code << np.uint64(1)) | ((a[:, i] >> b) & 1).astype(np.uint64)
order = np.argsort(code, kind="stable")
rank_morton = np.empty(m, np.int64); rank_morton[order] = np.arange(m)


def hilbert_code(a, bits):
    """Skilling's axes-to-transpose (Hilbert curve index) for the rows of a [m, d] non-negative int array."""
    X = [a[:, i].astype(np.int64).copy() for i in range(a.shape[1])]
    nd = len(X)
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(nd):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, nd):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[nd - 1] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(nd):
        X[i] ^= t
    h = np.zeros(a.shape[0], np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(nd):
            h = (h << np.uint64(1)) | ((X[i] >> b) & 1).astype(np.uint64)
    return h


horder = np.argsort(hilbert_code(a, bits), kind="stable")
rank_hilbert = np.empty(m, np.int64); rank_hilbert[horder] = np.arange(m)
# plain lexicographic order of a for comparison
lex = np.lexsort(tuple(a[:, i] for i in range(d - 1, -1, -1)))
rank_lex = np.empty(m, np.int64); rank_lex[lex] = np.arange(m)
print(f"ell={ell} m={m} bits/coord={bits}")
for name, rank in (("current (first touch)", np.arange(m)), ("morton(axis basis)", rank_morton), ("hilbert(axis basis)", rank_hilbert),
                   ("lexicographic(axis basis)", rank_lex)):
    rows = []
    for axis in range(d + 1):
        nb = nbr[axis].reshape(-1)
        src = np.tile(np.arange(m), 2)
        ok = nb >= 0
        dist = np.abs(rank[nb[ok]] - rank[src[ok]])
        rows.append([float((dist > t).mean()) for t in (1024, 32768, 87000, 262144, 1000000)] + [float(np.median(dist))])
    rows = np.array(rows)
    print(f"{name:28s} share of neighbours farther than 1k/32k/87k/262k/1M ids (mean over axes): "
          + " ".join(f"{v:.3f}" for v in rows[:, :5].mean(0)) + f"   median distance per axis: {rows[:, 5].astype(int).tolist()}")

# share of existing neighbours that fall into the same tile of T consecutive ids as their vertex (per axis, current ids):
# what an LDS-staged tile of rows would serve without a global gather
for T in (256, 512, 1024, 2048, 4096):
    shares = []
    for axis in range(d + 1):
        nb = nbr[axis].reshape(-1)
        src = np.tile(np.arange(m), 2)
        ok = nb >= 0
        shares.append(float((nb[ok] // T == src[ok] // T).mean()))
    print(f"tile {T:5d}: in-tile share per axis " + " ".join(f"{v:.2f}" for v in shares))
