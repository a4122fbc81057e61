"""CG / SLQ / marginal-likelihood harness on CPU (oracle-backed filter via the
reference's injection point), including BASELINE.json config 1: Snelson-1D,
RBFLattice(order=1) reaches the exact GP's train MLL within 0.1
(tests/train_snelson.py:96)."""
import os

import numpy as np
import pytest
import torch

import simplex_gp_amd as plx
from simplex_gp_amd import solvers
from oracle import oracle


def oracle_filter(src, ref, coeffs):
    return torch.from_numpy(oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy()))


@pytest.fixture
def cpu_method():
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    yield
    plx.LatticeFilterGeneral.method = None


def test_batched_cg_and_slq_on_dense_spd():
    g = torch.Generator().manual_seed(0)
    n = 120
    Q = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = Q @ Q.T / n + 0.5 * torch.eye(n, dtype=torch.float64)
    B = torch.randn(n, 4, generator=g, dtype=torch.float64)
    X, info = solvers.batched_cg(lambda V: A @ V, B, max_iter=500, tol=1e-10, want_tridiag=True)
    assert torch.allclose(X, torch.linalg.solve(A, B), atol=1e-7)
    assert info["iterations"] < 200 and info["tridiag"].shape[0] == 4
    # SLQ with many Rademacher probes approaches the exact log-determinant
    Z = (torch.randint(0, 2, (n, 64), generator=g).double() * 2 - 1)
    _, info = solvers.batched_cg(lambda V: A @ V, Z, max_iter=500, tol=1e-10, want_tridiag=True)
    est = float(solvers.slq_logdet(info["tridiag"], n))
    exact = float(torch.logdet(A))
    assert abs(est - exact) < 0.05 * abs(exact) + 1.0


def test_cg_iteration_floor_like_gpytorch():
    """With the reference's training tolerance (cg_tolerance(1.0), train_simplexgp.py:34) a CG that freezes a column
    as soon as its residual is below its right-hand side would stop after one step and hand SLQ a 1x1 tridiagonal.
    GPyTorch's linear_cg keeps iterating for min(10, max_iter-1) steps, 20 when tridiagonals are requested."""
    g = torch.Generator().manual_seed(3)
    n = 300
    Q = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = Q @ Q.T / n + 0.05 * torch.eye(n, dtype=torch.float64)
    Z = (torch.randint(0, 2, (n, 32), generator=g).double() * 2 - 1)
    exact = float(torch.logdet(A))
    _, info = solvers.batched_cg(lambda V: A @ V, Z, max_iter=500, tol=1.0, want_tridiag=True)
    assert info["iterations"] >= 20 and info["tridiag"].shape[-1] >= 20
    loose = float(solvers.slq_logdet(info["tridiag"], n))
    _, info_t = solvers.batched_cg(lambda V: A @ V, Z, max_iter=500, tol=1e-10, want_tridiag=True)
    tight = float(solvers.slq_logdet(info_t["tridiag"], n))
    assert abs(loose - tight) < 0.05 * abs(tight) and abs(tight - exact) < 0.1 * abs(exact)
    # without tridiagonals: 10 iterations at least; with the floor switched off the old one-step behaviour is back
    X, info = solvers.batched_cg(lambda V: A @ V, Z[:, :4], max_iter=500, tol=1.0)
    assert 10 <= info["iterations"] <= 12
    _, info0 = solvers.batched_cg(lambda V: A @ V, Z[:, :4], max_iter=500, tol=1.0, min_iter=0, check_every=1)
    assert info0["iterations"] <= 4
    # max_iter below the floor is respected; exactly converged columns freeze early instead of dividing 0 / 0
    _, info = solvers.batched_cg(lambda V: A @ V, Z[:, :4], max_iter=5, tol=1.0)
    assert info["iterations"] <= 5
    I = torch.eye(6, dtype=torch.float64)
    X, info = solvers.batched_cg(lambda V: 2.0 * V, I[:, :3].clone(), max_iter=50, tol=1e-3)
    assert torch.allclose(X, 0.5 * I[:, :3]) and torch.isfinite(X).all()


def test_mll_at_training_tolerance_is_close_to_tight_mll(cpu_method):
    """train/mll is computed at cg_tol = 1 (configs/simplexgp.yml): with the iteration floor it must agree with the
    tightly converged value, otherwise the training curve means nothing."""
    torch.manual_seed(0)
    n = 400
    x = torch.rand(n, 2) * 3
    y = torch.sin(2 * x[:, 0]) * torch.cos(x[:, 1]) + 0.1 * torch.randn(n)
    model = solvers.LatticeGP(plx.RBFLattice(order=1), min_noise=1e-2)
    with torch.no_grad():
        loose = float(solvers.marginal_log_likelihood(model, x, y, num_probes=20, cg_tol=1.0, max_cg_iter=500, seed=1))
        tight = float(solvers.marginal_log_likelihood(model, x, y, num_probes=20, cg_tol=1e-6, max_cg_iter=500, seed=1))
    assert abs(loose - tight) < 0.05 * abs(tight) + 0.02, (loose, tight)
    with pytest.raises(NotImplementedError):
        solvers.marginal_log_likelihood(model, x, y, reduce=lambda t: t)         # sharded MLL: not offered (docstring)


def _dense_spd(n, noise, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 1, generator=g, dtype=torch.float64) * 4
    K = torch.exp(-0.5 * torch.cdist(x / 0.5, x / 0.5).pow(2))
    return K, K + noise * torch.eye(n, dtype=torch.float64), g


def test_pivoted_cholesky_factor_and_woodbury():
    n, noise = 150, 1e-2
    K, A, g = _dense_spd(n, noise)
    pre = solvers.PivotedCholeskyPreconditioner(lambda V: K @ V, n, 1.0, noise, rank=25, device="cpu", dtype=torch.float64)
    # a smooth kernel matrix is numerically low rank: 25 pivots leave a small residual, and it shrinks with the rank
    err25 = float((K - pre.L @ pre.L.T).diagonal().sum())
    pre10 = solvers.PivotedCholeskyPreconditioner(lambda V: K @ V, n, 1.0, noise, rank=10, device="cpu", dtype=torch.float64)
    err10 = float((K - pre10.L @ pre10.L.T).diagonal().sum())
    assert 0 <= err25 < err10 < float(K.diagonal().sum()) and err25 < 1e-3 * n
    # full rank reproduces the matrix
    full = solvers.PivotedCholeskyPreconditioner(lambda V: K @ V, n, 1.0, noise, rank=n, device="cpu", dtype=torch.float64)
    assert float((K - full.L @ full.L.T).abs().max()) < 1e-6
    # Woodbury solve and the determinant lemma against the dense P
    Pd = pre.L @ pre.L.T + noise * torch.eye(n, dtype=torch.float64)
    R = torch.randn(n, 3, generator=g, dtype=torch.float64)
    assert torch.allclose(pre.solve(R), torch.linalg.solve(Pd, R), atol=1e-8, rtol=1e-6)
    assert abs(float(pre.logdet()) - float(torch.logdet(Pd))) < 1e-6 * n
    # samples have covariance P
    S = pre.sample(20000, generator=g)
    emp = S @ S.T / S.shape[1]
    assert float((emp - Pd).abs().max()) < 0.05


def test_preconditioned_cg_converges_faster_and_slq_matches_logdet():
    n, noise = 150, 1e-2
    K, A, g = _dense_spd(n, noise)
    B = torch.randn(n, 4, generator=g, dtype=torch.float64)
    exact = torch.linalg.solve(A, B)
    X0, info0 = solvers.batched_cg(lambda V: A @ V, B, max_iter=2000, tol=1e-8, check_every=1)
    pre = solvers.PivotedCholeskyPreconditioner(lambda V: K @ V, n, 1.0, noise, rank=25, device="cpu", dtype=torch.float64)
    X1, info1 = solvers.batched_cg(lambda V: A @ V, B, max_iter=2000, tol=1e-8, check_every=1, precond=pre)
    assert torch.allclose(X0, exact, atol=1e-5) and torch.allclose(X1, exact, atol=1e-5)
    assert info1["iterations"] * 3 < info0["iterations"], (info1["iterations"], info0["iterations"])
    # logdet A = logdet P + tr log(P^-1/2 A P^-1/2): the second term is small and its SLQ estimate has little
    # variance, so a few probes from N(0, P) beat many Rademacher probes without the preconditioner
    Z = pre.sample(16, generator=g)
    _, info = solvers.batched_cg(lambda V: A @ V, Z, max_iter=2000, tol=1e-10, want_tridiag=True, precond=pre)
    est = float(pre.logdet() + solvers.slq_logdet(info["tridiag"], n, weights=info["rz0"]))
    assert abs(est - float(torch.logdet(A))) < 0.5, (est, float(torch.logdet(A)))


def test_mll_with_preconditioner_matches_dense(cpu_method):
    torch.manual_seed(0)
    n = 60
    x = torch.randn(n, 2)
    y = torch.sin(x[:, 0]) + 0.1 * torch.randn(n)
    model = solvers.LatticeGP(plx.RBFLattice(order=1))
    with torch.no_grad():
        model.raw_noise.fill_(-3.0)               # small noise: the regime where the preconditioner matters
    mll = solvers.marginal_log_likelihood(model, x, y, num_probes=n, cg_tol=1e-7, seed=1, pre_size=15)
    mll.backward()
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    with torch.no_grad():
        Kd = model.kernel(x, x).evaluate()
        Kd = 0.5 * (Kd + Kd.T)
    s = model.raw_outputscale.detach().clone().requires_grad_(True)
    nz = model.raw_noise.detach().clone().requires_grad_(True)
    mu = model.mean.detach().clone().requires_grad_(True)
    Khat = torch.nn.functional.softplus(s) * Kd + (torch.nn.functional.softplus(nz) + model.min_noise) * torch.eye(n)
    r = (y - mu).reshape(-1, 1)
    dense = (-0.5 * (r * torch.linalg.solve(Khat, r)).sum() - 0.5 * torch.logdet(Khat) - 0.5 * n * np.log(2 * np.pi)) / n
    dense.backward()
    assert abs(float(mll) - float(dense)) < 0.05
    assert abs(float(got["mean"]) - float(mu.grad)) < 2e-2 * (1 + abs(float(mu.grad)))
    assert abs(float(got["raw_noise"]) - float(nz.grad)) < 5e-2 * (1 + abs(float(nz.grad)))
    assert abs(float(got["raw_outputscale"]) - float(s.grad)) < 5e-2 * (1 + abs(float(s.grad)))
    assert torch.isfinite(got["kernel.raw_lengthscale"]).all()


def test_mll_gradient_matches_dense_autograd(cpu_method):
    """Surrogate gradient of the CG/SLQ MLL vs autograd through a dense evaluation of the same operator."""
    torch.manual_seed(0)
    n = 60
    x = torch.randn(n, 2)
    y = torch.sin(x[:, 0]) + 0.1 * torch.randn(n)
    model = solvers.LatticeGP(plx.RBFLattice(order=1))
    mll = solvers.marginal_log_likelihood(model, x, y, num_probes=n, cg_tol=1e-7, seed=1)
    mll.backward()
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    # exact gradient of the quadratic-form part wrt mean / noise / outputscale using the dense lattice matrix
    with torch.no_grad():
        Kd = model.kernel(x, x).evaluate()
        Kd = 0.5 * (Kd + Kd.T)
    s = model.raw_outputscale.detach().clone().requires_grad_(True)
    nz = model.raw_noise.detach().clone().requires_grad_(True)
    mu = model.mean.detach().clone().requires_grad_(True)
    Khat = torch.nn.functional.softplus(s) * Kd + (torch.nn.functional.softplus(nz) + model.min_noise) * torch.eye(n)
    r = (y - mu).reshape(-1, 1)
    dense = (-0.5 * (r * torch.linalg.solve(Khat, r)).sum() - 0.5 * torch.logdet(Khat) - 0.5 * n * np.log(2 * np.pi)) / n
    dense.backward()
    assert abs(float(mll) - float(dense)) < 0.05
    assert abs(float(got["mean"]) - float(mu.grad)) < 2e-2 * (1 + abs(float(mu.grad)))
    assert abs(float(got["raw_noise"]) - float(nz.grad)) < 5e-2 * (1 + abs(float(nz.grad)))
    assert abs(float(got["raw_outputscale"]) - float(s.grad)) < 5e-2 * (1 + abs(float(s.grad)))
    assert torch.isfinite(got["kernel.raw_lengthscale"]).all()


def test_snelson_mll_matches_exact_gp(cpu_method, golden_dir):
    """Config 1 (tests/train_snelson.py): 100 Adam(lr=0.1) steps each, |MLL_lattice - MLL_exact| < 0.1."""
    sn = np.loadtxt(os.path.join(golden_dir, "snelson.csv"), delimiter=",", skiprows=1).astype(np.float32)
    x, y = torch.from_numpy(sn[:, :1].copy()), torch.from_numpy(sn[:, 1].copy())
    torch.manual_seed(0)
    exact = solvers.ExactRBFGP()
    opt = torch.optim.Adam(exact.parameters(), lr=0.1)
    for _ in range(100):
        opt.zero_grad()
        loss = -exact.mll(x, y)
        loss.backward()
        opt.step()
    exact_mll = float(exact.mll(x, y))

    model = solvers.LatticeGP(plx.RBFLattice(order=1))
    opt = torch.optim.Adam(model.parameters(), lr=0.1)
    for i in range(100):
        opt.zero_grad()
        mll = solvers.marginal_log_likelihood(model, x, y, num_probes=10, cg_tol=1e-4, max_cg_iter=500, seed=i)
        (-mll).backward()
        opt.step()
    with torch.no_grad():
        lattice_mll = float(solvers.marginal_log_likelihood(model, x, y, num_probes=50, cg_tol=1e-5, max_cg_iter=1000, seed=999))
    print("exact", exact_mll, "lattice", lattice_mll)
    assert abs(lattice_mll - exact_mll) < 0.1


def test_cap_host_threads_follows_the_cgroup_quota(monkeypatch, tmp_path):
    """solvers.cap_host_threads: the small host factorisations of a solve run under a thread pool no larger than half the
    cgroup CPU quota (a 128-thread pool under a 16-CPU quota stalls the whole process for the rest of the scheduler
    period); an explicit OMP_NUM_THREADS is left alone, and so is a process without a quota."""
    import builtins
    import torch
    from simplex_gp_amd import solvers
    before = torch.get_num_threads()
    real_open = builtins.open
    state = {"text": "400000 100000\n"}

    def fake_open(path, *a, **k):
        if str(path) == "/sys/fs/cgroup/cpu.max":
            import io
            return io.StringIO(state["text"])
        return real_open(path, *a, **k)

    try:
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
        monkeypatch.delenv("MKL_NUM_THREADS", raising=False)
        monkeypatch.setattr(builtins, "open", fake_open)
        torch.set_num_threads(8)
        assert solvers.cap_host_threads(force=True) == 2            # quota 4 CPUs -> 2 threads
        torch.set_num_threads(8)
        state["text"] = "max 100000\n"
        assert solvers.cap_host_threads(force=True) == 8            # no quota: nothing changes
        monkeypatch.setenv("OMP_NUM_THREADS", "8")
        state["text"] = "400000 100000\n"
        assert solvers.cap_host_threads(force=True) == 8            # the user has decided
        assert solvers.cap_host_threads() == 8                      # (once per process afterwards)
    finally:
        monkeypatch.setattr(builtins, "open", real_open)
        torch.set_num_threads(before)


def test_preconditioner_positions_key_notices_lengthscale_and_x_changes():
    """LatticeGP._positions_key / _same_positions (what lets khat_solve reuse the preconditioner's lattice): the key is
    the identity + version of x and of the raw lengthscale PARAMETER.  kernel.lengthscale is a fresh softplus output per
    access (version always 0), so a key built on it would keep matching after an optimizer step and CG would silently
    solve the old system."""
    import types
    model = solvers.LatticeGP(plx.RBFLattice(order=1))
    x = torch.randn(32, 2)
    pre = types.SimpleNamespace(ref_key=model._positions_key(x))
    assert model._same_positions(pre, x)
    assert model.kernel.lengthscale._version == 0
    model.kernel.lengthscale = 0.5                                   # in place on the raw parameter
    assert model.kernel.lengthscale._version == 0                    # ... which the softplus output does not show
    assert not model._same_positions(pre, x)
    pre.ref_key = model._positions_key(x)
    assert model._same_positions(pre, x)
    with torch.no_grad():
        model.kernel.raw_lengthscale.add_(0.1)                       # what an optimizer step does
    assert not model._same_positions(pre, x)
    pre.ref_key = model._positions_key(x)
    x2 = x.clone()
    assert not model._same_positions(pre, x2)                        # equal values, another tensor: not proven equal
    x.mul_(2.0)
    assert not model._same_positions(pre, x)                         # same tensor, changed in place
    pre.ref_key = model._positions_key(x)
    del x
    assert not model._same_positions(pre, x2)                        # the original is gone: nothing can match it
    pre.ref_key = None
    assert not model._same_positions(pre, x2)


def test_small_host_factorisations_restore_the_thread_count(monkeypatch):
    """The BLAS-pool cap applies for the duration of a small host factorisation and is undone afterwards: the library
    must not lower the host application's thread count for good (slq_terms / the preconditioner's k x k Cholesky)."""
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(4)
        monkeypatch.setattr(solvers, "_host_cap", 2)
        with solvers._small_host_factorisation():
            assert torch.get_num_threads() == 2
        assert torch.get_num_threads() == 4
        T = torch.diag_embed(torch.rand(3, 5) + 1.0)
        terms = solvers.slq_terms(T)
        assert torch.get_num_threads() == 4
        assert torch.allclose(terms.double(), T[:, 0, 0].log().double(), atol=1e-6)
        Tg = T.clone().requires_grad_(True)                           # a tridiagonal with a graph keeps the torch path
        solvers.slq_terms(Tg).sum().backward()
        assert Tg.grad is not None
    finally:
        torch.set_num_threads(before)
