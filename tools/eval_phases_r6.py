"""Round 6: where training.predict spends its time at the BASELINE.json configs[4] stand-in (N = 10,623, d = 18,
MaternLattice order 3, 5,976 held-out rows; the reference evaluates on the validation and test splits every epoch,
train_simplexgp.py:123-165).  Phases are timed with a device synchronisation on either side."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx                                              # noqa: E402
from simplex_gp_amd import solvers, training                              # noqa: E402


def main():
    n, d = 10623, 18
    if len(sys.argv) > 2:
        n, d = int(sys.argv[1]), int(sys.argv[2])
    ns = min(2656 + 3320, n // 2) if n < 100000 else n // 3
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).to(dev)
    xs = torch.randn(ns, d, generator=g).to(dev)
    y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).to(dev))
    kern = plx.RBFLattice(order=1, ard_num_dims=d) if "--rbf1" in sys.argv else plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d)
    model = solvers.LatticeGP(kern, min_noise=0.1 if "--rbf1" not in sys.argv else 1e-4).to(dev)
    solvers.cap_host_threads()

    def timed(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    for rep in range(3):
        with torch.no_grad():
            whole, t_all = timed(lambda: training.predict(model, x, y, xs, cg_tol=1e-2, lanc_iter=100, pre_size=100))
            r = (y - model.mean).reshape(-1, 1)
            K, t_k = timed(lambda: model.kernel(x, x))
            pre, t_pre = timed(lambda: model.preconditioner(x, 100, K=K))
            (alpha, info), t_cg = timed(lambda: model.khat_solve(x, r, K=K, max_iter=1000, tol=1e-2, precond=pre))
            K_star, t_ks = timed(lambda: model.kernel(xs, x))
            mean, t_mean = timed(lambda: K_star.matmul(alpha))
            (Q, T), t_lz = timed(lambda: training.lanczos(model.khat_matmul(x), r.squeeze(-1), 100))
            KQ, t_kq = timed(lambda: K_star.matmul(Q.contiguous()))
            _, t_chol = timed(lambda: torch.linalg.solve_triangular(
                torch.linalg.cholesky(T + 1e-6 * torch.eye(T.shape[0], device=dev)), KQ.t(), upper=False))
            _, t_prior = timed(lambda: model.kernel(xs, xs, diag=True))
        if rep == 2:
            rr = r.squeeze(-1)
            for label, gflag in (("eager caller-order", False), ("graph caller-order", True)):
                _, t = timed(lambda: training.lanczos(model.khat_matmul(x), rr, 100, graph=gflag))
                print(f"   lanczos {label}: {t:.2f} ms", flush=True)
            with model.khat_in_lattice_rows(x, K=K) as (mm_rows, to_rows, from_rows):
                rl = to_rows(r).squeeze(-1)
                for label, gflag in (("eager lattice-rows", False), ("graph lattice-rows", True)):
                    _, t = timed(lambda: training.lanczos(mm_rows, rl, 100, graph=gflag))
                    print(f"   lanczos {label}: {t:.2f} ms", flush=True)
                _, t = timed(lambda: [mm_rows(rl.reshape(-1, 1)) for _ in range(100)])
                print(f"   100 lattice-row MVMs alone: {t:.2f} ms", flush=True)
                sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                import bench
                print("   graph nodes of one MVM:", bench.graph_launches(lambda: mm_rows(rl.reshape(-1, 1))), flush=True)
                gq = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gq):
                    o = mm_rows(rl.reshape(-1, 1))
                _, t = timed(lambda: [gq.replay() for _ in range(100)])
                print(f"   100 replays of a one-MVM graph: {t:.2f} ms", flush=True)
            print("   refusals:", training._graph_refusals, flush=True)
        print(f"rep {rep}: predict {t_all:.2f} ms | kernel {t_k:.2f}  preconditioner {t_pre:.2f}  cg({info.get('iterations', '?')}) {t_cg:.2f}  "
              f"K_star {t_ks:.2f}  mean-mvm {t_mean:.2f}  lanczos(T {T.shape[0]}) {t_lz:.2f}  K_star@Q {t_kq:.2f}  chol+solve {t_chol:.2f}  "
              f"prior {t_prior:.2f}", flush=True)


if __name__ == "__main__":
    main()
