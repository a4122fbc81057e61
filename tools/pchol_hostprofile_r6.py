"""Round 6: the rank-100 factor at three small shapes under cProfile: where the HOST spends a build (read-backs, ctypes calls, the\nk x k Cholesky).  python tools/pchol_hostprofile_r6.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers
for n, d, kern in ((20000, 4, plx.MaternLattice(nu=1.5, order=2, ard_num_dims=4)), (10623, 18, plx.MaternLattice(nu=1.5, order=3, ard_num_dims=18)), (45730, 9, plx.RBFLattice(order=1, ard_num_dims=9))):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, d, generator=g).to(dev)
    model = solvers.LatticeGP(kern, min_noise=1e-2).to(dev)
    solvers.cap_host_threads()
    with torch.no_grad():
        pre = model.preconditioner(x, 100)
        s, noise = float(model.outputscale), float(model.noise)
        import cProfile, pstats
        for _ in range(3):
            p2 = solvers.LatticePreconditioner(pre.lat, s, noise, 100)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        for _ in range(10):
            p2 = solvers.LatticePreconditioner(pre.lat, s, noise, 100)
        torch.cuda.synchronize()
        pr.disable()
        print(f"n={n} d={d}: {(time.perf_counter()-t0)*100:.2f} ms per build, batches {p2.batches}")
        st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(12)
