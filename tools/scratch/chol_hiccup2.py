import time, torch, numpy as np, scipy.linalg as sl
from scipy.linalg import lapack
g = torch.Generator().manual_seed(0)
A = torch.randn(100, 300, generator=g, dtype=torch.float64)
C = (A @ A.T + torch.eye(100, dtype=torch.float64)).numpy()
x = torch.randn(4_000_000, device="cuda")
T = np.random.default_rng(0).standard_normal((11, 20, 20)); T = T @ T.transpose(0, 2, 1) + np.eye(20)
def run(label, fn, reps=400):
    ts = []
    for i in range(reps):
        y = (x * 2).sum(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(label, "median %.3f ms  p90 %.3f  max %.3f  >1ms: %d" % (np.median(ts), np.percentile(ts, 90), ts.max(), (ts > 1).sum()), flush=True)
def potri():
    c, info = lapack.dpotrf(C, lower=1)
    inv, info2 = lapack.dpotri(c, lower=1)
    return c, inv
run("scipy dpotrf + dpotri", potri)
def tri():
    L = np.linalg.cholesky(C)
    Li = sl.solve_triangular(L, np.eye(100), lower=True, check_finite=False)
    return Li.T @ Li
run("numpy cholesky + solve_triangular + matmul", tri)
run("numpy eigh 11x20x20", lambda: np.linalg.eigh(T))
c, inv = potri()
full = np.tril(inv) + np.tril(inv, -1).T
print("inverse error", np.abs(full @ C - np.eye(100)).max())
Tt = torch.from_numpy(T)
run("torch eigh 11x20x20 (default threads)", lambda: torch.linalg.eigh(Tt))
run("torch .log().sum() of 100", lambda: torch.from_numpy(C).diagonal().log().sum())
