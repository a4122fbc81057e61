"""Prediction / evaluation / training loop (simplex_gp_amd.training) on CPU via the oracle hook."""
import numpy as np
import pytest
import torch

import simplex_gp_amd as plx
from simplex_gp_amd import solvers, training
from oracle import oracle


def oracle_filter(src, ref, coeffs):
    return torch.from_numpy(oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy()))


@pytest.fixture
def cpu_method():
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    yield
    plx.LatticeFilterGeneral.method = None


def test_lanczos_reproduces_the_operator():
    g = torch.Generator().manual_seed(0)
    n = 40
    B = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = B @ B.T / n + torch.eye(n, dtype=torch.float64)
    Q, T = training.lanczos(lambda V: A @ V, torch.randn(n, generator=g, dtype=torch.float64), n)
    assert torch.allclose(Q.T @ Q, torch.eye(Q.shape[1], dtype=torch.float64), atol=1e-8)
    assert torch.allclose(Q.T @ A @ Q, T, atol=1e-8)


def test_lanczos_device_indexed_step_equals_the_eager_recurrence():
    """The step the HIP-graph form replays (training._lanczos_replayed: step index on the device, projection over the
    whole basis buffer, two Gram-Schmidt passes) run eagerly on the CPU: the same Q and T as the three-term loop, a
    breakdown found at the same place, and khat_in_lattice_rows away from the HIP path is the ordinary closure."""
    g = torch.Generator().manual_seed(1)
    n = 60
    B = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = B @ B.T / n + torch.eye(n, dtype=torch.float64)
    v0 = torch.randn(n, generator=g, dtype=torch.float64)
    for steps in (1, 2, 25, n):
        Q0, T0 = training.lanczos(lambda V: A @ V, v0, steps, graph=False)
        Q1, T1 = training._lanczos_replayed(lambda V: A @ V, v0, steps, check_every=8, capture=False)
        assert Q1.shape == Q0.shape and T1.shape == T0.shape
        assert torch.allclose(T1, T0, atol=1e-9) and torch.allclose(Q1, Q0, atol=1e-7)
        assert torch.allclose(Q1.T @ A @ Q1, T1, atol=1e-8)
    # an operator of rank 5 (+ identity): the Krylov space of v0 is exhausted after 6 steps; both forms cut there
    U = torch.randn(n, 5, generator=g, dtype=torch.float64)
    A5 = U @ U.T + torch.eye(n, dtype=torch.float64)
    Q0, T0 = training.lanczos(lambda V: A5 @ V, v0, 40, graph=False)
    Q1, T1 = training._lanczos_replayed(lambda V: A5 @ V, v0, 40, check_every=8, capture=False)
    assert Q0.shape[1] == Q1.shape[1] == 6 and torch.allclose(T1, T0, atol=1e-8)
    model = solvers.LatticeGP(plx.RBFLattice(order=1))
    x = torch.randn(10, 2)
    with model.khat_in_lattice_rows(x) as (mm, to_rows, from_rows):
        v = torch.randn(10, 1)
        assert to_rows(v) is v and from_rows(v) is v and callable(mm)


def test_predict_matches_dense_formulas(cpu_method):
    """Mean and (full-rank Lanczos) variance against the dense expressions built from the same lattice operators."""
    torch.manual_seed(0)
    n, ns = 60, 25
    x = torch.randn(n, 2)
    y = torch.sin(2 * x[:, 0]) + 0.1 * torch.randn(n)
    xs = torch.randn(ns, 2)
    model = solvers.LatticeGP(plx.RBFLattice(order=1), min_noise=1e-2)
    mean, var = training.predict(model, x, y, xs, cg_tol=1e-8, lanc_iter=n)
    with torch.no_grad():
        s, noise = model.outputscale, model.noise
        K = model.kernel(x, x).evaluate()
        Ks = model.kernel(xs, x).evaluate()              # [ns, n]
        Khat = s * K + noise * torch.eye(n)
        r = (y - model.mean).reshape(-1, 1)
        # the lattice operator is only approximately symmetric; CG / Lanczos see its action, so compare loosely
        mean_dense = model.mean + (s * Ks @ torch.linalg.solve(Khat, r)).squeeze(-1)
        var_dense = s - ((s * Ks) * torch.linalg.solve(Khat, (s * Ks).T).T).sum(1)
    assert mean.shape == (ns,) and var.shape == (ns,)
    assert torch.allclose(mean, mean_dense, atol=5e-2, rtol=5e-2)
    assert (var > 0).all() and torch.allclose(var, var_dense.clamp_min(1e-8), atol=0.1)


def test_early_stopper_contract():
    st = training.EarlyStopper(patience=2, delta=0.1)
    st(1.0, "a")
    assert st.info() == "a" and not st.is_done()
    st(1.05, "b")                       # not more than delta better: a miss
    assert st.info() == "a"
    st(1.2, "c")                        # improvement resets the counter
    assert st.info() == "c" and not st.is_done()
    st(1.0, "d")
    st(1.1, "e")
    assert st.is_done() and st.info() == "c"
    with pytest.raises(AssertionError):
        st(5.0, "f")
    assert not training.EarlyStopper(patience=-1)._misses and not training.EarlyStopper(patience=-1).is_done()


def test_fit_improves_validation_rmse(cpu_method, tmp_path):
    g = torch.Generator().manual_seed(1)
    x = torch.rand(400, 2, generator=g) * 4 - 2
    f = torch.sin(2 * x[:, 0]) * torch.cos(x[:, 1])
    y = f + 0.1 * torch.randn(400, generator=g)
    tr, va, te = slice(0, 250), slice(250, 325), slice(325, 400)
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=2), min_noise=1e-3)
    before = training.evaluate(model, x[tr], y[tr], x[va], y[va], label="val")
    ckpt = tmp_path / "model.pt"
    history, best = training.fit(model, (x[tr], y[tr]), val=(x[va], y[va]), test=(x[te], y[te]), epochs=25, lr=0.1,
                                 patience=50, cg_tol=1e-2, checkpoint=str(ckpt))
    assert len(history) == 25 and best is not None and ckpt.exists()
    assert best["summary"]["val/rmse"] <= before["val/rmse"] + 1e-6
    assert best["summary"]["val/rmse"] < 0.6 * float(y[va].std())        # far better than predicting the mean
    assert np.isfinite(best["summary"]["test/nll"]) and best["summary"]["test/rmse"] < 0.5
    assert history[-1]["train/mll"] > history[0]["train/mll"]
    state = torch.load(str(ckpt))
    assert set(state) == set(model.state_dict())


def test_fit_evaluates_both_splits_from_one_cache(cpu_method, monkeypatch):
    """The loop's evaluation (train_simplexgp.py:123-165: validation, then test, after every step) solves and runs Lanczos
    ONCE per evaluated epoch -- GPyTorch's eval mode keeps its mean / variance caches between the reference's two test()
    calls -- and both splits' metrics are those of predict() called split by split."""
    torch.manual_seed(1)
    n = 80
    x = torch.randn(n, 2)
    y = torch.sin(2 * x[:, 0]) + 0.1 * torch.randn(n)
    tr, va, te = slice(0, 50), slice(50, 65), slice(65, 80)
    model = solvers.LatticeGP(plx.RBFLattice(order=1), min_noise=1e-2)
    made = []
    real = training.PredictionCache

    class Counting(real):
        def __init__(self, *a, **k):
            made.append(1)
            super().__init__(*a, **k)
    monkeypatch.setattr(training, "PredictionCache", Counting)
    hist, best = training.fit(model, (x[tr], y[tr]), val=(x[va], y[va]), test=(x[te], y[te]), epochs=3, lr=0.05, num_probes=4,
                              cg_tol=1e-4, cg_eval_tol=1e-6, lanc_iter=50, pre_size=0)
    assert len(made) == 3 and len(hist) == 3 and {"val/rmse", "val/nll", "test/rmse", "test/mae"} <= set(hist[-1])
    monkeypatch.setattr(training, "PredictionCache", real)
    want_v = training.evaluate(model, x[tr], y[tr], x[va], y[va], label="val", cg_tol=1e-6, lanc_iter=50, pre_size=0)
    want_t = training.evaluate(model, x[tr], y[tr], x[te], y[te], label="test", cg_tol=1e-6, lanc_iter=50, pre_size=0)
    for k, v in {**want_v, **want_t}.items():
        assert abs(hist[-1][k] - v) <= 1e-5 * max(1.0, abs(v)), k
    # a cache handed to predict() is used as it is
    cache = real(model, x[tr], y[tr], cg_tol=1e-6, lanc_iter=50, pre_size=0)
    m1, v1 = training.predict(model, x[tr], y[tr], x[va], cache=cache)
    m2, v2 = training.predict(model, x[tr], y[tr], x[va], cg_tol=1e-6, lanc_iter=50, pre_size=0)
    assert torch.allclose(m1, m2, atol=1e-6) and torch.allclose(v1, v2, atol=1e-6)
