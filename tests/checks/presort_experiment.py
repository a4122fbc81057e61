#!/usr/bin/env python3
"""Experiment: how much do the apply kernels gain if points arrive spatially sorted?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from tools.archive.ab_apply import timeit, RBF1

def morton(q, bits):
    n, d = q.shape
    code = np.zeros(n, np.uint64)
    for b in range(bits - 1, -1, -1):
        for j in range(d):
            code = (code << np.uint64(1)) | ((q[:, j] >> np.uint64(b)) & np.uint64(1))
    return code

def orders(x, ell):
    r = x / ell
    out = {"original": np.arange(len(x))}
    # lattice-coordinate key: rounded zero-colour vertex of every point (from the CPU oracle)
    from oracle import oracle
    oracle.set_exact_mode(False)
    o = oracle.Lattice(r.astype(np.float32), RBF1)
    gr = o.greedy.astype(np.int64) // (x.shape[1] + 1)
    gr -= gr.min(0)
    out["lex greedy"] = np.lexsort(gr.T[::-1])
    ev = o.entry_vertex
    out["by corner-0 vertex id"] = np.argsort(ev[:, 0], kind="stable")
    o.close(); oracle.set_exact_mode(True)
    for cell in (0.25, 0.5, 1.0):
        q = np.floor(r / cell).astype(np.int64)
        q -= q.min(0)
        out[f"lex cell={cell}"] = np.lexsort(q.T[::-1])
        bits = int(np.ceil(np.log2(q.max() + 1)))
        if bits * x.shape[1] <= 64:
            out[f"morton cell={cell}"] = np.argsort(morton(q.astype(np.uint64), bits), kind="stable")
    return out

def main():
    n, d = 1_000_000, 8
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).numpy()
    v = torch.randn(n, 1, generator=g)
    for ell in (1.0, 0.6931):
        for name, perm in orders(x, ell).items():
            xs = torch.from_numpy(np.ascontiguousarray(x[perm] / ell)).cuda()
            vs = v[perm].contiguous().cuda()
            lat = plx.Lattice().build(xs, RBF1)
            vals, scratch, out = lat.new_values(1), lat.new_values(1), torch.empty_like(vs)
            ts = timeit(lambda: lat.splat(vs, vals)); tb = timeit(lambda: lat.blur(vals, scratch)); tl = timeit(lambda: lat.slice(vals, out))
            ta = timeit(lambda: lat.apply(vs, out))
            lat.set_timing(True); lat.build(xs, RBF1); bt = lat.build_times_ms(); lat.set_timing(False)
            print(f"ell={ell} {name:18s} m={lat.m} splat {ts:7.2f} blur {tb:7.2f} slice {tl:7.2f} apply {ta:7.2f} us | build " + " ".join(f"{k}={t:.3f}" for k, t in bt.items()), flush=True)
            lat.close()

if __name__ == "__main__":
    main()
