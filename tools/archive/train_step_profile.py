#!/usr/bin/env python3
"""Where does one marginal-likelihood training step go?  N=1e6, d=8, 10 probes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).cuda())
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
opt = torch.optim.Adam(model.parameters(), lr=0.05)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for step in range(4):
    opt.zero_grad()
    t0 = sync()
    mll = solvers.marginal_log_likelihood(model, x, y, num_probes=10, cg_tol=1.0, max_cg_iter=50, seed=step)
    t1 = sync()
    (-mll).backward()
    t2 = sync()
    opt.step()
    t3 = sync()
    print(json.dumps({"step": step, "forward_s": round(t1 - t0, 4), "backward_s": round(t2 - t1, 4), "optim_s": round(t3 - t2, 4),
                      "cg_iterations": mll.cg_info["iterations"], "mll": round(float(mll.detach()), 4),
                      "peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}), flush=True)
