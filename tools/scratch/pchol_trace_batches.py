import sys, time, torch
sys.path.insert(0, "/root/repo")
import simplex_gp_amd as plx
from simplex_gp_amd import solvers
n, d, rank = 1_000_000, 8, 100
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
orig = torch.Tensor.tolist
log = []
def spy(self):
    r = orig(self)
    if self.dtype == torch.int32 and self.numel() == 3:
        log.append(r)
    return r
torch.Tensor.tolist = spy
with torch.no_grad():
    K = model.kernel(x, x)
    pre = model.preconditioner(x, rank, K=K)
    print("batches", pre.batches, "planned", pre.planned_batches, log)
    for ell in (1.5, 0.4):
        log.clear()
        model.kernel.lengthscale = ell
        pre = model.preconditioner(x, rank)
        print("ell", ell, "m", pre.lat.m, "batches", pre.batches, "planned", pre.planned_batches, "sparse", pre.sparse_batches, log)
