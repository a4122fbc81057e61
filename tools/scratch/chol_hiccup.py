import time, torch, numpy as np
print("threads", torch.get_num_threads(), torch.__config__.parallel_info()[:300])
g = torch.Generator().manual_seed(0)
A = torch.randn(100, 300, generator=g, dtype=torch.float64)
C = A @ A.T + torch.eye(100, dtype=torch.float64)
x = torch.randn(4_000_000, device="cuda")
def run(label, fn, reps=300):
    ts = []
    for i in range(reps):
        y = (x * 2).sum(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(label, "median %.3f ms  p90 %.3f  max %.3f  >1ms: %d" % (np.median(ts), np.percentile(ts, 90), ts.max(), (ts > 1).sum()), flush=True)
run("torch.linalg.cholesky (default threads)", lambda: torch.linalg.cholesky(C))
run("torch cholesky + cholesky_inverse", lambda: torch.cholesky_inverse(torch.linalg.cholesky(C)))
Cn = C.numpy()
run("numpy cholesky", lambda: np.linalg.cholesky(Cn))
run("numpy inv", lambda: np.linalg.inv(Cn))
n0 = torch.get_num_threads()
def one():
    torch.set_num_threads(1)
    try:
        return torch.cholesky_inverse(torch.linalg.cholesky(C))
    finally:
        torch.set_num_threads(n0)
run("torch, 1 thread around the call", one)
torch.set_num_threads(1)
run("torch, 1 thread for good", lambda: torch.cholesky_inverse(torch.linalg.cholesky(C)))
