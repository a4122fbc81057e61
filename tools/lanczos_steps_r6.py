"""Round 6: `steps` Lanczos steps on a cheap symmetric operator (diagonal + rank 3) at n rows: the re-orthogonalisation
kernels by themselves (for a kernel trace), native (plx_lanczos_step) and torch forms timed in one process.
    python tools/lanczos_steps_r6.py [n] [steps] [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplex_gp_amd import training                                       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator().manual_seed(0)
dvec = (1.0 + 3.0 * torch.rand(n, generator=g)).cuda()
U = (torch.randn(n, 3, generator=g) / n ** 0.5).cuda()
v0 = torch.randn(n, generator=g).cuda()


def mm(V):
    return dvec[:, None] * V + U @ (U.T @ V)


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


best = {}
for r in range(rounds + 1):
    for label, kw in (("native", {}), ("torch eager", {"graph": False})):
        t = timed(lambda: training.lanczos(mm, v0, steps, **kw))
        if r:
            best[label] = min(best.get(label, 1e9), t)
t_mm = min(timed(lambda: [mm(v0[:, None]) for _ in range(steps)]) for _ in range(3))
print(f"n = {n}, {steps} steps: " + ", ".join(f"{k} {v:.2f} ms" for k, v in best.items()) + f"; the {steps} MVMs alone {t_mm:.2f} ms", flush=True)
