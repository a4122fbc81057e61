"""Prediction, evaluation and the training loop around the lattice operator
(SURVEY 8f-3): what experiments/train_simplexgp.py does with GPyTorch, written
against simplex_gp_amd.solvers so that it runs where GPyTorch is absent.

    predict     mean through the rectangular operator K(x*, X) (py:142-160) and a
                Lanczos (LOVE-style) variance cache, as `fast_pred_var` does
                (train_simplexgp.py:63-72)
    evaluate    RMSE / MAE / NLL exactly as train_simplexgp.py:74-84 defines them
    EarlyStopper  same contract as experiments/utils.py:170-199
    fit         Adam on the CG/SLQ marginal likelihood with validation-RMSE early
                stopping and a best-state checkpoint (train_simplexgp.py:117-165)
"""
import math

import torch

from .solvers import LatticeGP, marginal_log_likelihood


def lanczos(matmul, v0, steps, check_every=8):
    """`steps` Lanczos iterations with full re-orthogonalisation.
    Returns Q [n, t] (orthonormal) and the tridiagonal T [t, t] with Q^T A Q = T.

    The basis lives in one preallocated [steps, n] buffer (stacking the vectors anew in every step copied O(steps^2 n) bytes)
    and the breakdown test -- beta below 1e-6 |alpha_0|: the Krylov space is exhausted -- reads the device every
    `check_every` steps instead of twice per step (a step is one MVM of ~80 us at the elevators size: the two host
    synchronisations cost more than the step); vectors produced after a breakdown are discarded when it is found."""
    n = v0.shape[0]
    Qb = torch.zeros(steps, n, dtype=v0.dtype, device=v0.device)
    alphas = torch.zeros(steps, dtype=v0.dtype, device=v0.device)
    betas = torch.zeros(steps, dtype=v0.dtype, device=v0.device)
    Qb[0] = v0 / v0.norm()
    t = steps

    def first_breakdown(upto):
        small = betas[:upto] < 1e-6 * alphas[0].abs()
        return int(torch.nonzero(small)[0]) + 1 if bool(small.any()) else None

    for i in range(steps):
        w = matmul(Qb[i].unsqueeze(-1)).squeeze(-1)
        if i > 0:
            w = w - betas[i - 1] * Qb[i - 1]
        alpha = torch.dot(w, Qb[i])
        w = w - alpha * Qb[i]
        Qi = Qb[:i + 1]
        w = w - Qi.t() @ (Qi @ w)                 # full re-orthogonalisation
        alphas[i] = alpha
        if i + 1 == steps:
            break
        beta = w.norm()
        betas[i] = beta
        Qb[i + 1] = w / beta.clamp_min(1e-30)
        if (i + 1) % check_every == 0:
            cut = first_breakdown(i + 1)
            if cut is not None:
                t = cut
                break
    else:
        pass
    if t == steps and steps > 1:
        cut = first_breakdown(steps - 1)
        if cut is not None:
            t = cut
    T = torch.diag(alphas[:t])
    if t > 1:
        idx = torch.arange(t - 1, device=v0.device)
        T[idx, idx + 1] = betas[:t - 1]
        T[idx + 1, idx] = betas[:t - 1]
    return Qb[:t].t(), T


@torch.no_grad()
def predict(model, x, y, x_star, max_cg_iter=1000, cg_tol=1e-2, lanc_iter=100, variance=True, pre_size=100):
    """Posterior mean and variance of the latent function at x_star.

    mean = mu + s K(x*, X) (s K + sigma^2 I)^-1 (y - mu)          one CG solve + one rectangular MVM
    var  = s k(x*, x*) - || L^-1 Q^T (s K(X, x*)) ||^2            K^-1 ~ Q T^-1 Q^T from `lanc_iter`
           Lanczos steps started at y - mu, T = L L^T; one rectangular MVM with lanc_iter columns
    pre_size: rank of the pivoted-Cholesky preconditioner of the mean solve (the reference's test() default,
    train_simplexgp.py:60: 100; 0 = plain CG)
    """
    assert isinstance(model, LatticeGP)
    r = (y - model.mean).reshape(-1, 1)
    K = model.kernel(x, x)                                # one operator (one lattice) for the preconditioner and the solve
    precond = model.preconditioner(x, pre_size, K=K) if pre_size > 0 else None
    alpha, _ = model.khat_solve(x, r, K=K, max_iter=max_cg_iter, tol=cg_tol, precond=precond)
    K_star = model.kernel(x_star, x)                      # RectangularLazyLattice, [n*, n]
    s = model.outputscale
    mean = model.mean + s * K_star.matmul(alpha).squeeze(-1)
    if not variance:
        return mean, None
    Q, T = lanczos(model.khat_matmul(x), r.squeeze(-1), min(lanc_iter, x.shape[0]))
    jitter = 1e-6 * T.diagonal().abs().max()
    Lc = torch.linalg.cholesky(T + jitter * torch.eye(T.shape[0], dtype=T.dtype, device=T.device))
    KQ = s * K_star.matmul(Q.contiguous())                # [n*, t]
    proj = torch.linalg.solve_triangular(Lc, KQ.t(), upper=False)
    prior = s * model.kernel(x_star, x_star, diag=True)
    var = (prior - (proj ** 2).sum(0)).clamp_min(1e-8)
    return mean, var


@torch.no_grad()
def evaluate(model, x, y, x_star, y_star, label="test", **predict_args):
    """RMSE, MAE and mean negative log predictive density (train_simplexgp.py:74-84)."""
    mean, var = predict(model, x, y, x_star, **predict_args)
    rmse = (mean - y_star).pow(2).mean(0).sqrt()
    mae = (mean - y_star).abs().mean(0)
    nll = -torch.distributions.Normal(mean, (var + model.noise).sqrt()).log_prob(y_star).mean()
    return {f"{label}/rmse": rmse.item(), f"{label}/mae": mae.item(), f"{label}/nll": nll.item()}


class EarlyStopper:
    """Keeps the best (score, info); done after `patience` non-improving calls
    (improvement = more than `delta` above the best score)."""

    def __init__(self, patience=10, delta=1e-4):
        self.patience, self.delta = patience, delta
        self._misses = 0
        self._best_score = None
        self._best_info = None

    def is_done(self):
        return self.patience >= 0 and self._misses >= self.patience

    def info(self):
        return self._best_info

    def __call__(self, score, info):
        assert not self.is_done()
        if self._best_score is None or score >= self._best_score + self.delta:
            self._best_score, self._best_info, self._misses = score, info, 0
        else:
            self._misses += 1


def make_optimizer(model, lr=0.1):
    """torch.optim.Adam(model.parameters(), lr) as the reference builds it (train_simplexgp.py:84), with the update of all
    parameters in ONE launch when they live on the GPU (Adam's `fused` implementation: the same update rule; the default
    per-parameter loop is a dozen launches and a third of a millisecond of Python for five scalars)."""
    params = list(model.parameters())
    if params and all(p.is_cuda and p.is_floating_point() for p in params):
        try:
            return torch.optim.Adam(params, lr=lr, fused=True)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.Adam(params, lr=lr)


def fit(model, train, val=None, test=None, epochs=100, lr=0.1, patience=200, log_every=1, num_probes=10,
        cg_iter=1000, cg_tol=1.0, cg_eval_tol=1e-2, lanc_iter=100, pre_size=100, checkpoint=None, log=None,
        cap_host_threads=False):
    """Adam on -MLL; every `log_every` epochs evaluate on val/test, keep the state
    with the best validation RMSE, stop after `patience` evaluations without
    improvement; optionally torch.save the best state_dict to `checkpoint`.
    Defaults are the reference's (train_simplexgp.py:87-90): `pre_size` = 100 is the rank of the pivoted-Cholesky
    preconditioner (train_simplexgp.py:36; on the HIP path it is built and applied natively, solvers.LatticePreconditioner;
    0 = plain CG).
    cap_host_threads=True: the process's host BLAS pool is capped at half the container's CPU quota for good
    (solvers.cap_host_threads: a 128-thread pool under a 16-CPU quota stalls the loop's small host steps for 30-100 ms
    every few epochs).  Opt-in, because it changes the host application's thread count; the library itself only caps
    the pool for the duration of its own small host factorisations and restores it."""
    x, y = train
    if cap_host_threads:
        from . import solvers
        solvers.cap_host_threads()
    opt = make_optimizer(model, lr=lr)
    stopper = EarlyStopper(patience=patience)
    history = []
    for epoch in range(epochs):
        opt.zero_grad()
        mll = marginal_log_likelihood(model, x, y, num_probes=num_probes, max_cg_iter=cg_iter, cg_tol=cg_tol, seed=epoch,
                                      pre_size=pre_size)
        (-mll).backward()
        opt.step()
        row = {"epoch": epoch + 1, "train/mll": float(mll.detach())}
        if val is not None and epoch % log_every == 0:
            row.update(evaluate(model, x, y, val[0], val[1], label="val", max_cg_iter=cg_iter, cg_tol=cg_eval_tol,
                                lanc_iter=lanc_iter, pre_size=pre_size))
            if test is not None:
                row.update(evaluate(model, x, y, test[0], test[1], label="test", max_cg_iter=cg_iter,
                                    cg_tol=cg_eval_tol, lanc_iter=lanc_iter, pre_size=pre_size))
            stopper(-row["val/rmse"], {"state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()},
                                       "summary": dict(row)})
            if checkpoint is not None:
                torch.save(stopper.info()["state_dict"], checkpoint)
        history.append(row)
        if log is not None:
            log(row)
        if val is not None and stopper.is_done():
            break
    return history, (stopper.info() if val is not None else None)
