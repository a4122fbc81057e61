#!/usr/bin/env python3
"""Round 5 CPU study (numpy only, no GPU, no oracle): which cache lines do the neighbour lookups of the lattice build
touch, per wave of 64 consecutive vertices, under different slot functions of the vertex table?

Background: neighbor_kernel at N = 1e6, d = 8, l = 0.25 moves 15.6 GB to write a 643 MB table.
Every random 4-byte read costs one 128-byte line fill, so what matters is the number of DISTINCT lines a wave touches.
Variants:
  random        the shipped table: slot = mix(key) & mask
  block<L>      locality-preserving slot function: slot = (H(a >> s) << L) | low bits of the blur-axis coordinates a,
                L low bits in all (one per coordinate for the first L coordinates): the +1 neighbour along a "low-bit" axis
                stays inside the 2^L-slot block of its vertex half of the time
  xcd-sliced    the shipped table + occupancy bitmap, but every lookup is served by the XCD that owns the slice of the
                bitmap its slot falls in, so that the bitmap reads are L2 hits (counted as zero line fills)

Prints, per variant: line fills per lookup for the bitmap / table / key-compare accesses, probe lengths, and the implied
traffic at 128 B per fill.

    python tools/slot_locality_study.py [n] [ell]
"""
import sys
import numpy as np

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ell = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
d = 8
D1 = d + 1
TAPS = np.array([0.34608543, 1.0, 0.34608543], np.float32)


def synth(n, d, seed=1234):
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g).numpy()


def embed(x):
    """SURVEY appendix A steps 1-6 in numpy (fp32): greedy, rank per point."""
    mom0 = TAPS.sum(); mom1 = (np.arange(3, dtype=np.float32) * TAPS).sum(); mom2 = (np.arange(3, dtype=np.float32) ** 2 * TAPS).sum()
    var = np.float32(mom2 / mom0 - (mom1 / mom0) ** 2)
    sf = np.array([np.float32(D1) * np.sqrt(var + np.float32(1 / 6)) / np.sqrt(np.float32((i + 1) * (i + 2))) for i in range(d)], np.float32)
    el = np.zeros((x.shape[0], D1), np.float32)
    el[:, d] = np.float32(-d) * x[:, d - 1] * sf[d - 1]
    for i in range(d - 1, 0, -1):
        el[:, i] = el[:, i + 1] - np.float32(i) * x[:, i - 1] * sf[i - 1] + np.float32(i + 2) * x[:, i] * sf[i]
    el[:, 0] = el[:, 1] + np.float32(2) * x[:, 0] * sf[0]
    v = el / np.float32(D1)
    up = np.ceil(v) * D1; down = np.floor(v) * D1
    gr = np.where(up - el < el - down, up, down).astype(np.int32)
    s = (gr.sum(1).astype(np.float32) * np.float32(1 / D1)).astype(np.int32)
    diff = el - gr
    rk = np.zeros_like(gr)
    for i in range(d):
        for j in range(i + 1, D1):
            lt = diff[:, i] < diff[:, j]
            rk[:, i] += lt; rk[:, j] += ~lt
    pos = s > 0; neg = s < 0
    sc = s[:, None]
    wrap_p = pos[:, None] & (rk >= D1 - sc)
    wrap_n = neg[:, None] & (rk < -sc)
    gr = gr - wrap_p * D1 + wrap_n * D1
    rk = rk + np.where(pos[:, None], np.where(wrap_p, sc - D1, sc), 0) + np.where(neg[:, None], np.where(wrap_n, D1 + sc, sc), 0)
    return gr, rk, np.rint(v).astype(np.int32)


def morton(q):
    q = q - q.min(0)
    bits = int(np.ceil(np.log2(q.max() + 1)))
    code = np.zeros(q.shape[0], np.uint64)
    for b in range(bits - 1, -1, -1):
        for c in range(q.shape[1]):
            code = (code << np.uint64(1)) | ((q[:, c] >> b) & 1).astype(np.uint64)
    return code


def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def lines_per_wave(addr_bytes, valid, wave=64, line=128):
    """sum over waves of the number of distinct lines among the valid accesses"""
    ln = (addr_bytes // line).astype(np.int64)
    ln = np.where(valid, ln, -1)
    pad = (-len(ln)) % wave
    ln = np.concatenate([ln, np.full(pad, -1, np.int64)]).reshape(-1, wave)
    ln.sort(axis=1)
    distinct = (ln[:, 1:] != ln[:, :-1]) & (ln[:, 1:] >= 0)
    return int(distinct.sum() + (ln[:, 0] >= 0).sum())


x = synth(n, d) / np.float32(ell)
gr, rk, q = embed(x.astype(np.float32))
order = np.argsort(morton(q), kind="stable")          # lattice order of the points (Z-curve of the rounded coordinates)
gr, rk = gr[order], rk[order]
# corner keys (first d coordinates) in entry order e = p * D1 + r
r = np.arange(D1)[None, :, None]
keys = gr[:, None, :d] + np.where(rk[:, None, :d] <= d - r, r, r - D1)        # [n, D1, d]
keys = keys.reshape(-1, d)
kd = -keys.sum(1)
a = (kd[:, None] - keys) // D1                                                  # blur-axis coordinates of every corner
amin = a.min(0)
au = (a - amin).astype(np.uint64)
width = [int(np.ceil(np.log2(au[:, c].max() + 1))) for c in range(d)]
code = np.zeros(len(au), np.uint64)
for c in range(d):
    code = (code << np.uint64(width[c])) | au[:, c]
assert sum(width) <= 64
uniq, first, inv = np.unique(code, return_index=True, return_inverse=True)
ft_order = np.argsort(first, kind="stable")                                     # first-touch numbering (h:73-79)
m = len(uniq)
vid_of_uniq = np.empty(m, np.int64); vid_of_uniq[ft_order] = np.arange(m)
va = a[first[ft_order]]                                                         # [m, d] coordinates in vertex id order
vcode = uniq[ft_order]
print(f"n = {n}, l = {ell}: m = {m} ({m / (n * D1):.3f} of the corners), code bits {sum(width)}")
cap = 1024
while cap < 2 * n * D1:
    cap *= 2
sorted_codes = uniq                                                             # for membership tests


def member(acoords):
    ok = np.all((acoords >= amin) & (acoords - amin < (1 << np.array(width))), axis=1)
    c = np.zeros(len(acoords), np.uint64)
    aa = np.where(ok[:, None], acoords - amin, 0).astype(np.uint64)
    for k in range(d):
        c = (c << np.uint64(width[k])) | aa[:, k]
    pos = np.searchsorted(sorted_codes, c)
    pos = np.minimum(pos, m - 1)
    hit = ok & (sorted_codes[pos] == c)
    return hit, vid_of_uniq[pos]


M = np.array([0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F, 0x165667B1, 0xD3A2646D, 0xFD7046C5, 0xB55A4F09], np.uint64)


def slot_random(acoords):
    s = (acoords.astype(np.int64).astype(np.uint64) * M[None, :]).sum(1) & np.uint64(0xFFFFFFFF)
    return mix32(s) & np.uint64(cap - 1)


def make_slot_block(L, capb):
    sh = np.array([1 if c < L else 0 for c in range(d)])

    def f(acoords):
        hi = acoords >> sh
        lo = np.zeros(len(acoords), np.uint64)
        for c in range(L):
            lo = (lo << np.uint64(1)) | (acoords[:, c] & 1).astype(np.uint64)
        s = (hi.astype(np.int64).astype(np.uint64) * M[None, :]).sum(1) & np.uint64(0xFFFFFFFF)
        return (((mix32(s) << np.uint64(L)) | lo) & np.uint64(capb - 1))
    return f


def build_table(slot_fn, capt):
    """linear probing, insertion in vertex id order; returns table (slot -> vertex or -1) and vslot"""
    table = np.full(capt, -1, np.int64)
    h = slot_fn(va).astype(np.int64)
    pending = np.arange(m)
    probes = np.zeros(m, np.int64)
    while len(pending):
        hh = h[pending]
        # the lowest pending id per free slot wins this round
        free = table[hh] < 0
        cand = pending[free]
        ch = hh[free]
        o = np.argsort(ch, kind="stable")
        ch_s, cand_s = ch[o], cand[o]
        win = np.ones(len(ch_s), bool); win[1:] = ch_s[1:] != ch_s[:-1]
        table[ch_s[win]] = cand_s[win]
        placed = np.zeros(m, bool); placed[cand_s[win]] = True
        rest = pending[~placed[pending]]
        h[rest] = (h[rest] + 1) % capt
        probes[rest] += 1
        pending = rest
    return table, h, probes


def study(name, slot_fn, capt, sliced_bitmap=False, bitmap=True):
    table, vslot, ins_probes = build_table(slot_fn, capt)
    occ = table >= 0
    tot = dict(bitmap=0, table=0, keys=0, lookups=0, present=0, probes=0)
    for axis in range(D1):
        na = va.copy()
        if axis < d:
            na[:, axis] += -1          # tap nid = +1: a_axis - 1  (neighbor_kernel: ajn = aj - nid)
        else:
            na += 1                    # axis d moves every coordinate: key[c] - nid for all c -> a_c = (kd' - kc')/(d+1) with kd' = kd + d*nid ...
        hit, nid = member(na)
        h = slot_fn(na).astype(np.int64)
        active = np.ones(m, bool)
        step = 0
        while active.any():
            idx = np.nonzero(active)[0]
            hh = h[idx]
            # 1. bitmap (one bit per slot)
            if bitmap and not sliced_bitmap:
                tot["bitmap"] += lines_per_wave(np.where(active, h // 8, 0), active)
            o = occ[hh]
            # 2. table word of occupied slots
            act2 = np.zeros(m, bool); act2[idx[o]] = True
            if not bitmap:
                tot["table"] += lines_per_wave(np.where(active, h * 4, 0), active)
            else:
                tot["table"] += lines_per_wave(np.where(act2, h * 4, 0), act2)
            # 3. key compare with the vertex found there
            v = np.where(act2, table[np.where(act2, h, 0)], 0)
            tot["keys"] += lines_per_wave(v * 16, act2)
            tot["probes"] += int(active.sum())
            match = act2 & (v == np.where(hit, nid, -2))
            active = act2 & ~match
            h = np.where(active, (h + 1) % capt, h)
            step += 1
        tot["lookups"] += m
        tot["present"] += int(hit.sum())
    L = tot["lookups"]
    fills = tot["bitmap"] + tot["table"] + tot["keys"]
    print(f"{name:28s} cap {capt:>11d} load {m / capt:.3f} insert probes/vertex {ins_probes.mean():.3f} | lookups {L} present {tot['present'] / L:.3f} "
          f"probes/lookup {tot['probes'] / L:.3f} | line fills per lookup: bitmap {tot['bitmap'] / L:.3f} table {tot['table'] / L:.3f} "
          f"keys {tot['keys'] / L:.3f} total {fills / L:.3f} -> {fills * 128 / 1e9:.2f} GB", flush=True)


study("random, no bitmap", slot_random, cap, bitmap=False)
study("random + bitmap (shipped)", slot_random, cap)
study("random + XCD-sliced bitmap", slot_random, cap, sliced_bitmap=True)
for L in (3, 5):
    for mult in (1, 4):
        study(f"block{L} x{mult} + bitmap", make_slot_block(L, cap * mult), cap * mult)
        study(f"block{L} x{mult} + sliced bitmap", make_slot_block(L, cap * mult), cap * mult, sliced_bitmap=True)
