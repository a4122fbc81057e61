#!/usr/bin/env python3
"""Randomised check of plx_lanczos_step (through training.lanczos) against the same recurrence in fp64 on the host of the
same operator: symmetric positive operators A = diag + low rank of random size -- every row-span shape of the kernels and their
boundaries (n = 1, 2, 63 .. 65, 255 .. 257, 65,536 +- 1, 262,144 +- 1, 1,048,576 +- 1, 2,097,152 and one row more: served by
the torch form) --, 1 .. 256 steps, Krylov spaces that are exhausted early (identity + rank <= 3: cut after at most rank + 2).
Checked: T against the fp64 recurrence's (leading block, where two fp32 runs still agree), Q orthonormal, Q^T A Q = T,
bit-identical repeats.
    python tests/checks/fuzz_lanczos.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplex_gp_amd import training                                       # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EDGES = [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1025, 65535, 65536, 65537, 262143, 262144, 262145, 1048575, 1048576, 1048577,
         2097152, 2097153]
worst_t = worst_q = worst_h = 0.0
for c in range(cases):
    n = EDGES[c] if c < len(EDGES) else int(rng.choice([rng.integers(1, 300), rng.integers(300, 70000), rng.integers(70000, 400000)]))
    big = n > 400000
    steps = int(min(n, rng.integers(1, 40 if big else 257)))
    rank = int(rng.integers(0, 4))
    flat = bool(rng.integers(0, 4) == 0) and rank > 0 and n > 8            # identity + low rank: the Krylov space has rank + 1 vectors
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    dvec = torch.ones(n) if flat else 1.0 + 3.0 * torch.rand(n, generator=g)
    U = torch.randn(n, max(rank, 1), generator=g) / max(n, 1) ** 0.5 * (1.0 if rank else 0.0)
    v0 = torch.randn(n, generator=g)
    dv, Ud, v0d = dvec.cuda(), U.cuda(), v0.cuda()

    def mm(V):
        return dv[:, None] * V + Ud @ (Ud.T @ V)
    Q, T = training.lanczos(mm, v0d, steps)
    Q2, T2 = training.lanczos(mm, v0d, steps)
    assert torch.equal(Q, Q2) and torch.equal(T, T2), ("not deterministic", c, n, steps)
    t = T.shape[0]
    # the fp64 recurrence on the host (three-term + one full pass), cut where it breaks down
    dv64, U64 = dvec.double(), U.double()
    mm64 = lambda v: dv64 * v + U64 @ (U64.T @ v)                          # noqa: E731
    Qh = torch.zeros(steps + 1, n, dtype=torch.float64)
    Qh[0] = v0.double() / v0.double().norm()
    al, be, t64 = [], [], steps
    for i in range(steps):
        w = mm64(Qh[i])
        if i:
            w = w - be[-1] * Qh[i - 1]
        a = float(w @ Qh[i])
        w = w - a * Qh[i]
        w = w - Qh[:i + 1].T @ (Qh[:i + 1] @ w)
        al.append(a)
        b = float(w.norm())
        if i + 1 < steps and b < 1e-6 * abs(al[0]):
            t64 = i + 1
            break
        be.append(b)
        Qh[i + 1] = w / max(b, 1e-300)
    if flat:
        # (three nearly equal eigenvalues above a flat spectrum: the recurrence is ill-determined past its second vector, in
        # any precision -- only the cut and the invariants below are compared)
        assert t <= min(steps, rank + 2), ("cut", c, n, steps, t, t64, rank)
    else:
        assert t == t64 == steps, ("cut", c, n, steps, t, t64)
    k = 1 if flat else min(t, t64, 10)
    scale = max(abs(x) for x in al[:k])
    Tk = T[:k, :k].double().cpu()
    dt = max(float((Tk.diagonal() - torch.tensor(al[:k])).abs().max()),
             float((Tk.diagonal(1) - torch.tensor(be[:k - 1])).abs().max()) if k > 1 else 0.0) / scale
    dq = float((Q.T @ Q - torch.eye(t, device="cuda")).abs().max())
    dh = float((Q.T @ mm(Q.contiguous()) - T).abs().max()) / scale
    worst_t, worst_q, worst_h = max(worst_t, dt), max(worst_q, dq), max(worst_h, dh)
    assert dt <= 2e-4 and dq <= 2e-4 and dh <= 5e-4, (c, n, steps, t, t64, flat, dt, dq, dh)
    if c % 10 == 0:
        print(f"case {c}: n = {n}, steps = {steps}, t = {t} (fp64: {t64}){' flat' if flat else ''}: T {dt:.1e}  Q^T Q - I {dq:.1e}  Q^T A Q - T {dh:.1e}", flush=True)
print(f"{cases} cases: worst |T - T_fp64| / |T| (leading 10) {worst_t:.2e}, worst |Q^T Q - I| {worst_q:.2e}, worst |Q^T A Q - T| / |T| {worst_h:.2e}")
