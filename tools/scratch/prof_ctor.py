import cProfile, pstats, sys, os, time, io
sys.path.insert(0, "/root/repo")
import torch
import simplex_gp_amd as plx
from simplex_gp_amd import solvers
n, d, rank = 1_000_000, 8, 100
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
with torch.no_grad():
    K = model.kernel(x, x)
    pre0 = model.preconditioner(x, rank, K=K)
    lat0 = pre0.lat
    for rep in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pr = cProfile.Profile(); pr.enable()
        pre = solvers.LatticePreconditioner(lat0, float(model.outputscale), float(model.noise), rank, sparse_rows=True)
        torch.cuda.synchronize()
        pr.disable()
        ms = (time.perf_counter() - t0) * 1e3
        print("ctor ms", ms)
        if ms > 20:
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(6); print(s.getvalue()[:1500])
