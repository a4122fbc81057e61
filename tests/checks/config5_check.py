#!/usr/bin/env python3
"""BASELINE.json config 5 stand-in (UCI elevators is not available): N=10,623, d=18, MaternLattice(nu=1.5,
order=3): parity of one MVM against the CPU oracle, MVM timing, and a few MLL training steps."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers
from oracle import oracle
from tools.archive.ab_apply import timeit

n, d = 10623, 18
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g); y = torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g)
k = plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d)
taps = k.dkernel_fn.get_coeffs().numpy()
ref = (x / 1.0).contiguous()
v = torch.randn(n, 1, generator=g)
lat = plx.Lattice(); lat.set_timing(True); lat.build(ref.cuda(), taps); bt = lat.build_times_ms(); lat.set_timing(False)
out = lat.apply(v.cuda())
oracle.set_exact_mode(False); want, m = oracle.filter(v.numpy(), ref.numpy(), taps, return_m=True); oracle.set_exact_mode(True)
err = float(np.linalg.norm(out.cpu().numpy() - want) / np.linalg.norm(want))
t_apply = timeit(lambda: lat.apply(v.cuda()))
vc = v.cuda(); o = torch.empty_like(vc)
t_apply = timeit(lambda: lat.apply(vc, o))
t0 = time.perf_counter(); oracle.filter(v.numpy(), ref.numpy(), taps); t_cpu = time.perf_counter() - t0
model = solvers.LatticeGP(k, min_noise=0.1).cuda()
opt = torch.optim.Adam(model.parameters(), lr=0.1)
xc, yc = x.cuda(), y.cuda()
torch.cuda.synchronize(); t0 = time.perf_counter(); mlls = []
for i in range(5):
    opt.zero_grad()
    mll = solvers.marginal_log_likelihood(model, xc, yc, num_probes=10, cg_tol=1.0, max_cg_iter=500, seed=i)
    (-mll).backward(); opt.step(); mlls.append(float(mll.detach()))
torch.cuda.synchronize(); t_train = (time.perf_counter() - t0) / 5
print(json.dumps({"config": "N=10623 d=18 Matern-1.5 order 3 (stand-in for elevators)", "m": lat.m, "m_oracle": m,
                  "rel_l2_vs_oracle": err, "apply_us": round(t_apply, 1), "cpu_oracle_s": round(t_cpu, 3),
                  "build_ms": {a: round(b, 3) for a, b in bt.items()}, "train_step_s": round(t_train, 3), "mll": mlls}))
