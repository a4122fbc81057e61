#!/usr/bin/env python3
"""CPU study (oracle only, no GPU): how many staging rows do block / tile structures need?

Point blocks (slice_block / splat_block): B consecutive points in lattice order, rows = distinct (block, vertex) pairs.
Vertex tiles (the transpose): T consecutive vertices in Morton order, rows = distinct (tile, point) pairs.
Prints rows / nnz for several sizes.  Usage: tile_sharing_study.py [n] [ell]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ell = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
d = 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).numpy() / ell
oracle.set_exact_mode(False)
lat = oracle.Lattice(x, np.array([0.34608543, 1.0, 0.34608543], np.float32))
m = lat.m
ev = lat.entry_vertex.astype(np.int64)          # [n, d+1]
keys = lat.keys.astype(np.int64)                # [m, d]
greedy = lat.greedy.astype(np.int64)            # [n, d+1]
print(f"n={n} ell={ell} m={m} nnz={n*(d+1)}")


def interleave(a, bits):
    code = np.zeros(a.shape[0], np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(a.shape[1]):
            code = (code << np.uint64(1)) | ((a[:, i] >> b) & 1).astype(np.uint64)
    return code


# lattice point order: Z-curve of the rounded lattice coordinates (first ncoord that fit 64 bits)
q = greedy // (d + 1)
q = q - q.min(0)
bits = int(np.ceil(np.log2(q.max() + 1)))
ncoord = min(d + 1, 64 // bits)
porder = np.argsort(interleave(q[:, :ncoord], bits), kind="stable")
ev = ev[porder]
# first-touch renumbering along that order, then Morton renumbering of the vertices
kd = -keys.sum(1)
a = (kd[:, None] - keys) // (d + 1)
a -= a.min(0)
vb = int(np.ceil(np.log2(a.max() + 1)))
vorder = np.argsort(interleave(a, vb), kind="stable")
vrank = np.empty(m, np.int64); vrank[vorder] = np.arange(m)
evm = vrank[ev]                                 # Morton vertex id of every corner, points in lattice order
nnz = evm.size
pid = np.repeat(np.arange(n), d + 1).reshape(n, d + 1)

for B in (224, 448, 896, 1820):
    rows = np.unique((pid // B) * m + evm).size
    print(f"point blocks of {B:5d}: rows = {rows:9d} = {rows/nnz:.3f} nnz  ({nnz/rows:.2f} corners/row)")
for T in (64, 128, 256, 512, 1024, 4096):
    rows = np.unique((evm // T) * n + pid).size
    print(f"vertex tiles of {T:5d}: rows = {rows:9d} = {rows/nnz:.3f} nnz  ({nnz/rows:.2f} corners/row, {nnz/((m+T-1)//T):.0f} corners/tile)")
# first-touch vertex numbering for comparison
_, first = np.unique(ev.reshape(-1), return_index=True)
ft = np.empty(m, np.int64); ft[ev.reshape(-1)[np.sort(first)]] = np.arange(m)
evf = ft[ev]
for T in (256, 1024):
    rows = np.unique((evf // T) * n + pid).size
    print(f"first-touch vertex tiles of {T:5d}: rows = {rows:9d} = {rows/nnz:.3f} nnz")
# (block, tile) segments: how many distinct vertex tiles does a point block touch?
for B, T in ((448, 256), (448, 1024)):
    seg = np.unique((pid // B) * (m // T + 1) + evm // T).size
    print(f"(block {B}, tile {T}) segments: {seg}  = {seg/((n+B-1)//B):.1f} per block")

# hashed insert, one wave = corner plane r of 64 consecutive points: how many distinct keys does a wave probe
# (a) as it is (lanes equal to their left neighbour copy its slot), (b) with a full in-wave de-duplication,
# (c) if a wave took ALL d+1 corners of 7 consecutive points (63 lanes) and de-duplicated those
W = 64
nw = n // W
evw = evm[: nw * W].reshape(nw, W, d + 1)
adj = (evw[:, 1:, :] != evw[:, :-1, :]).sum() + nw * (d + 1)
full = sum(len(np.unique(evw[i, :, r])) for i in range(0, nw, 50) for r in range(d + 1)) * 50 / 1.0
print(f"probes per corner: adjacent-lane dedupe {adj / (nw * W * (d + 1)):.3f}, full in-wave dedupe {full / (nw * W * (d + 1)):.3f}")
g7 = evm[: (n // 7) * 7].reshape(-1, 7 * (d + 1))
c7 = sum(len(np.unique(g7[i])) for i in range(0, g7.shape[0], 50)) * 50
print(f"all corners of 7 consecutive points per wave, de-duplicated: {c7 / (g7.shape[0] * 7 * (d + 1)):.3f} probes per corner")
g28 = evm[: (n // 28) * 28].reshape(-1, 28 * (d + 1))
c28 = sum(len(np.unique(g28[i])) for i in range(0, g28.shape[0], 50)) * 50
print(f"all corners of 28 consecutive points per workgroup (252 threads), de-duplicated: {c28 / (g28.shape[0] * 28 * (d + 1)):.3f} probes per corner")

# per-block range of (Morton) vertex ids: how many key bits would a per-block sort need after subtracting the block minimum?
for B in (448, 672):
    nb = n // B
    blk = evm[: nb * B].reshape(nb, B * (d + 1))
    span = blk.max(1) - blk.min(1)
    bits = np.ceil(np.log2(span + 2)).astype(int)
    print(f"blocks of {B} points: vertex-id span bits: median {int(np.median(bits))}, 90 % <= {int(np.percentile(bits, 90))}, max {bits.max()} (m needs {int(np.ceil(np.log2(m)))}); "
          f"share of blocks needing <= 16 bits: {(bits <= 16).mean():.2f}, <= 12: {(bits <= 12).mean():.2f}")
