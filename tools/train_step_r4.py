#!/usr/bin/env python3
"""Round 4: one marginal-likelihood training step as the reference runs it (experiments/train_simplexgp.py:29-57:
10 probes, cg_tolerance(1.0), max_preconditioner_size(pre_size)), split by phase.  N=1e6, d=8, RBF order 1 by default.

    python tools/train_step_r4.py [--n N] [--d D] [--pre 0 100] [--steps 3] [--matern]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers
solvers.cap_host_threads()      # the process-wide BLAS-pool cap is the caller's opt-in (bench.py does the same)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--pre", type=int, nargs="*", default=[0, 100])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--max-cg", type=int, default=500)
ap.add_argument("--matern", action="store_true")
ap.add_argument("--no-profile", action="store_true", help="no phase synchronisation: whole-step wall time only")
args = ap.parse_args()
n, d = args.n, args.d
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).cuda())
def sync(): torch.cuda.synchronize(); return time.perf_counter()
def throttled():
    """(periods throttled, microseconds throttled) of this container's CPU quota so far (cgroup v2), or None"""
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(st["nr_throttled"]), int(st["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None
for pre in args.pre:
    kern = plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d) if args.matern else plx.RBFLattice(order=1, ard_num_dims=d)
    model = solvers.LatticeGP(kern).cuda()
    from simplex_gp_amd import training
    opt = training.make_optimizer(model, lr=0.1)
    for step in range(args.steps):
        opt.zero_grad()
        prof = None if args.no_profile else {}
        th0 = throttled()
        rs0 = torch.cuda.memory_reserved()
        t0 = sync()
        mll = solvers.marginal_log_likelihood(model, x, y, num_probes=10, cg_tol=1.0, max_cg_iter=args.max_cg, seed=step,
                                              pre_size=pre, profile=prof)
        t1 = sync()
        (-mll).backward()
        t2 = sync()
        opt.step()
        t3 = sync()
        row = {"pre_size": pre, "step": step, "forward_ms": round((t1 - t0) * 1e3, 2), "backward_ms": round((t2 - t1) * 1e3, 2),
               "optim_ms": round((t3 - t2) * 1e3, 2), "step_ms": round((t3 - t0) * 1e3, 2),
               "cg_iterations": mll.cg_info["iterations"], "mll": round(float(mll.detach()), 4),
               "peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
        if prof is not None:
            row["phases_ms"] = {k: round(v, 2) for k, v in prof.items()}
        th1 = throttled()
        if th0 and th1:
            row["cpu_throttled"] = {"periods": th1[0] - th0[0], "ms": round((th1[1] - th0[1]) / 1e3, 1)}     # the container's CPU quota ran out during the step
        row["reserved_grew_MB"] = round((torch.cuda.memory_reserved() - rs0) / 1e6, 1)                     # torch's allocator asked the driver for memory
        row["lattices_m"] = [e[0].m for e in plx.lattice_cache()._entries.values()][-2:]      # the last two lattices in the cache (RBF: the previous step's and this step's; a profile whose derivative taps differ builds two per step)
        print(json.dumps(row), flush=True)
    plx.lattice_cache().clear()
