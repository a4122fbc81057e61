"""Round 6: 3 x 100 Lanczos steps in lattice row order at the configs[4] stand-in (or `n d [--rbf1]`), nothing else (for a
kernel trace); --torch: the torch-op recurrence instead of plx_lanczos_step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers, training                              # noqa: E402

n, d = 10623, 18
if len(sys.argv) > 2:
    n, d = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).to(dev)
r = torch.randn(n, 1, generator=g).to(dev)
kern = plx.RBFLattice(order=1, ard_num_dims=d) if "--rbf1" in sys.argv else plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d)
model = solvers.LatticeGP(kern, min_noise=0.1).to(dev)
with torch.no_grad(), model.khat_in_lattice_rows(x) as (mm_rows, to_rows, from_rows):
    for _ in range(3):
        training.lanczos(mm_rows, to_rows(r).squeeze(-1), 100, **({"graph": False} if "--torch" in sys.argv else {}))
torch.cuda.synchronize()
