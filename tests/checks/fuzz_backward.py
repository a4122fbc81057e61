#!/usr/bin/env python3
"""Randomised check of LatticeFilterGeneral's gradients on the HIP path (fused plx_apply_backward where the shape
allows, the three-call native form otherwise) against the reference formulation (py:113-123) evaluated over the CPU
oracle filter.  RBF orders 1-3 and Matern-1.5 order 3 (forward and derivative taps differ there)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from oracle import oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)

def oracle_filter(src, ref, coeffs):
    oracle.set_exact_mode(False)
    try:
        return torch.from_numpy(oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy()))
    finally:
        oracle.set_exact_mode(True)

kernels = [plx.DiscretizedKernelFN(plx.rbf, 1), plx.DiscretizedKernelFN(plx.rbf, 2), plx.DiscretizedKernelFN(plx.rbf, 3),
           plx.DiscretizedKernelFN(lambda d2: plx.Matern.apply(d2, 1.5), 3)]
worst = (0.0, None)
nfused = ncancel = 0
worst_terms = 0.0
for c in range(cases):
    n = int(rng.choice([1, 5, 64, 300, 1025, 3000]))
    d = int(rng.integers(1, 13))
    L = int(rng.choice([1, 2, 3, 7, 8, 11, 16, 20, 28]))
    dk = kernels[int(rng.integers(0, len(kernels)))]
    scale = float(rng.choice([0.3, 1.0, 3.0]))
    x0 = torch.from_numpy((rng.standard_normal((n, d)) * scale).astype(np.float32))
    v0 = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32))
    w0 = torch.from_numpy(rng.standard_normal((n, L)).astype(np.float32))
    only_x = bool(rng.integers(0, 2))
    def grads(device):
        x = x0.clone().to(device).requires_grad_(True)
        v = v0.clone().to(device).requires_grad_(not only_x)
        out = plx.LatticeFilterGeneral.apply(v, x, dk)
        (out * w0.to(device)).sum().backward()
        return (None if only_x else v.grad.cpu().numpy()), x.grad.cpu().numpy(), out.detach().cpu().numpy()
    plx.LatticeFilterGeneral.method = None
    got = grads("cuda")
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    try:
        want = grads("cpu")
    finally:
        plx.LatticeFilterGeneral.method = None
    nfused += int(plx.Lattice.backward_fusable(L, d))
    for name, a, b in (("grad_src", got[0], want[0]), ("grad_x", got[1], want[1]), ("out", got[2], want[2])):
        if a is None:
            continue
        err = float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-20))
        if err > worst[0] and err <= 1e-4:
            worst = (err, (name, n, d, L, c))
        if name == "grad_x" and err > 1e-4:
            # py:122 is a difference of products, -2 sum(sx*wg - src*wgx + gx*ws - g*wsx): where points are isolated
            # (wgx = x*wg exactly) the true gradient is 0 and both results are rounding noise.  Judge the absolute
            # error against the size of the terms instead.
            wg = oracle_filter(w0, x0.detach(), dk.get_deriv_coeffs())
            terms = 2.0 * (v0.abs() * wg.abs()).sum(1, keepdim=True) * x0.detach().abs()
            rel_terms = float(np.linalg.norm(a.astype(np.float64) - b) / max(float(terms.norm()), 1e-20))
            print(f"case {c}: grad_x rel {err:.2e} but {rel_terms:.2e} of the cancelling terms", (n, d, L, scale), flush=True)
            if rel_terms <= 1e-5:
                ncancel += 1
                worst_terms = max(worst_terms, rel_terms)
                continue
        if err > 1e-4 or not np.isfinite(a).all():
            print("FAIL", name, err, (n, d, L, scale, c), flush=True)
            sys.exit(1)
print(f"{cases} cases ({nfused} through the fused kernel): worst rel-L2 {worst[0]:.2e} at {worst[1]}; {ncancel} grad_x cases with a "
      f"vanishing true gradient judged against their cancelling terms, worst {worst_terms:.2e} of the terms")
