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
import torch

from .solvers import LatticeGP, marginal_log_likelihood


LANCZOS_GRAPH = "auto"        # "auto": replay one captured step on a GPU when the loop is launch-bound (below); False: never
LANCZOS_GRAPH_MIN_STEPS = 16
LANCZOS_GRAPH_MAX_ROWS = 65536   # the replayed step projects on the WHOLE basis buffer: at N = 1e6 that is 2 x 2 passes over
#                                  400 MB per step (measured: 100 steps 53.6 ms replayed, 30.0 ms eager); at N = 10,623 the
#                                  buffer is 4 MB and both forms take 14.3 ms -- the replay's gain there is that the loop no
#                                  longer depends on the host keeping ahead of 40 launches per step (caller-order operator
#                                  through autograd wrappers: 69 ms eager, 16.4 ms replayed)
_graph_refusals = []          # why a capture was refused (kept for the tests and for a curious user; at most 8 entries)


def _tridiagonal(alphas, betas, t):
    T = torch.diag(alphas[:t])
    if t > 1:
        idx = torch.arange(t - 1, device=alphas.device)
        T[idx, idx + 1] = betas[:t - 1]
        T[idx + 1, idx] = betas[:t - 1]
    return T


def _first_breakdown(alphas, betas, upto):
    """1 + the index of the first beta below 1e-6 |alpha_0| among the first `upto` (the Krylov space is exhausted
    there), or None.  One device read."""
    small = betas[:upto] < 1e-6 * alphas[0].abs()
    return int(torch.nonzero(small)[0]) + 1 if bool(small.any()) else None


def _lanczos_replayed(matmul, v0, steps, check_every, capture=True):
    """The Lanczos recurrence as ONE captured HIP graph replayed once per step.  At the sizes where the variance cache
    matters (N ~ 1e4: the reference's UCI sets) a step of the torch-op recurrence is ~40 launches of a few microseconds of
    work each (100 steps at N = 10,623, d = 18: 16-27 ms through the caller-order operator); the guide's prescription for a
    launch-bound inner loop is a graph.  Measured (profiles/r06_measured.md section 8): the replay is as fast as the eager
    loop when the DEVICE is the bound (3.6 us per dependent launch either way) and 4x faster when the host is; the
    default on a GPU is the native step (_lanczos_native: fewer launches), this form serves operators in other dtypes.  For the graph to be the
    same in every step, the step index lives on the device: row `idx` of the basis is read with index_select, row
    idx + 1 written with index_copy_, and the projection runs over ALL rows of the basis buffer (rows not yet written are
    zero and contribute nothing) -- two classical Gram-Schmidt passes (which include the alpha q_i and beta q_{i-1}
    terms of the three-term recurrence: alpha_i is the q_i entry of the projection).  The first step runs eagerly
    (it sizes every buffer the MVM asks the library for; a capture refuses allocations), the second is captured,
    the rest are replays.  Breakdown is looked for every `check_every` replays.
    capture=False runs the same device-indexed step eagerly (any device: what the CPU tests check the formulation with)."""
    n, dev, dt = v0.shape[0], v0.device, v0.dtype
    Qb = torch.zeros(steps + 1, n, dtype=dt, device=dev)          # one spare row: the last step writes q_steps
    alphas = torch.zeros(steps, dtype=dt, device=dev)
    betas = torch.zeros(steps, dtype=dt, device=dev)
    idx = torch.zeros(1, dtype=torch.long, device=dev)
    Qb[0] = v0 / v0.norm()

    def step():
        w = matmul(Qb.index_select(0, idx).reshape(n, 1)).reshape(n)
        c = torch.mv(Qb, w)
        w = torch.addmv(w, Qb.t(), c, alpha=-1.0)
        c2 = torch.mv(Qb, w)
        w = torch.addmv(w, Qb.t(), c2, alpha=-1.0)
        alphas.index_copy_(0, idx, (c + c2).index_select(0, idx))
        beta = torch.linalg.vector_norm(w).reshape(1)
        betas.index_copy_(0, idx, beta)
        idx.add_(1)
        Qb.index_copy_(0, idx, (w / beta.clamp_min(1e-30)).unsqueeze(0))

    step()
    done = 1
    if steps > 1:
        replay = step
        if capture:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):               # (capturing records the step without running it)
                step()
            replay = graph.replay
        t = None
        while done < steps and t is None:
            for _ in range(min(check_every, steps - done)):
                replay()
                done += 1
            if done < steps:
                t = _first_breakdown(alphas, betas, done)
    t = _first_breakdown(alphas, betas, min(done, steps - 1)) if steps > 1 else None
    t = steps if t is None else t
    return Qb[:t].t(), _tridiagonal(alphas, betas, t)


LANCZOS_NATIVE = True         # on a GPU in fp32: re-orthogonalisation, coefficients and the next basis vector as four launches
#                               of libplx (plx_lanczos_step) instead of ~19 torch launches per step


def _lanczos_native(matmul, v0, steps, check_every):
    """The recurrence with everything but the MVM in plx_lanczos_step (csrc/plx_lanczos.hip): per step the operator's own
    launches + 4.  The components along q_{i-1}, q_i first (the three-term recurrence), then one classical Gram-Schmidt
    pass against the whole basis built so far -- the torch form's order, which is what keeps the pass stable;
    deterministic.  None when the library does not serve this shape
    (more than 256 steps or 2,097,152 rows)."""
    import ctypes
    from . import _native as nv
    lib = nv.lib()
    n, dev = v0.shape[0], v0.device
    work_floats = int(lib.plx_lanczos_work_floats(n))
    if steps > int(lib.plx_lanczos_max_rows()) or work_floats < 0:
        return None
    ld = (n + 63) // 64 * 64                                                 # rows of the basis start on 256-byte boundaries
    Qb = torch.zeros(steps + 1, ld, dtype=torch.float32, device=dev)         # one spare row: the last step writes q_steps
    alphas = torch.zeros(steps, dtype=torch.float32, device=dev)
    betas = torch.zeros(steps, dtype=torch.float32, device=dev)
    work = torch.empty(work_floats, dtype=torch.float32, device=dev)
    Qb[0, :n] = v0 / v0.norm()
    t = None
    with torch.cuda.device(dev):
        for i in range(steps):
            w = matmul(Qb[i, :n].unsqueeze(-1)).reshape(n)
            if not w.is_contiguous() or w.dtype != torch.float32 or w.data_ptr() == Qb[i].data_ptr():
                w = w.to(torch.float32).contiguous().clone()                 # (the step overwrites w: never the basis row itself)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            nv.check(lib.plx_lanczos_step(ctypes.c_void_p(Qb.data_ptr()), ld, ctypes.c_void_p(w.data_ptr()), n, i,
                                          ctypes.c_void_p(alphas.data_ptr()), ctypes.c_void_p(betas.data_ptr()),
                                          ctypes.c_void_p(work.data_ptr()), stream), "plx_lanczos_step")
            if (i + 1) % check_every == 0 and i + 1 < steps:
                t = _first_breakdown(alphas, betas, i + 1)
                if t is not None:
                    break
    if t is None and steps > 1:
        t = _first_breakdown(alphas, betas, steps - 1)
    t = steps if t is None else t
    return Qb[:t, :n].t(), _tridiagonal(alphas, betas, t)


def lanczos(matmul, v0, steps, check_every=8, graph=None, native=None):
    """`steps` Lanczos iterations with full re-orthogonalisation.
    Returns Q [n, t] (orthonormal) and the tridiagonal T [t, t] with Q^T A Q = T.

    On a GPU in fp32 (native, default LANCZOS_NATIVE): everything around the MVM is plx_lanczos_step -- four launches per
    step (_lanczos_native).  Otherwise torch ops: the basis lives in one preallocated [steps, n] buffer (stacking the
    vectors anew in every step copied O(steps^2 n) bytes) and the breakdown test -- beta below 1e-6 |alpha_0|: the
    Krylov space is exhausted -- reads the device every `check_every` steps instead of twice per step; vectors produced
    after a breakdown are discarded when it is found.
    graph (default LANCZOS_GRAPH; looked at when the native step is off or does not serve the shape): the torch step
    captured once into a HIP graph and replayed (_lanczos_replayed); `matmul` must then be capturable -- no host
    read-backs, no allocations outside torch's pool: the operators of this package are, once their lattice is built.  A
    refused capture falls back to the eager loop (the reason is kept in _graph_refusals)."""
    if native is None:
        native = LANCZOS_NATIVE and graph is None                   # (an explicit graph=True / False asks for a torch form)
    if native and v0.is_cuda and v0.dtype == torch.float32 and not torch.cuda.is_current_stream_capturing():
        out = _lanczos_native(matmul, v0, steps, max(int(check_every), 32))
        if out is not None:
            return out
    graph = LANCZOS_GRAPH if graph is None else graph
    if graph and v0.is_cuda and (graph is True or (steps >= LANCZOS_GRAPH_MIN_STEPS and v0.shape[0] <= LANCZOS_GRAPH_MAX_ROWS)) \
            and not torch.cuda.is_current_stream_capturing():
        try:
            return _lanczos_replayed(matmul, v0, steps, max(int(check_every), 32))
        except RuntimeError as e:
            if graph is True:
                raise
            if len(_graph_refusals) < 8:
                _graph_refusals.append(f"{type(e).__name__}: {e}")
            torch.cuda.synchronize(v0.device)
    n = v0.shape[0]
    Qb = torch.zeros(steps, n, dtype=v0.dtype, device=v0.device)
    alphas = torch.zeros(steps, dtype=v0.dtype, device=v0.device)
    betas = torch.zeros(steps, dtype=v0.dtype, device=v0.device)
    Qb[0] = v0 / v0.norm()
    t = steps
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
            cut = _first_breakdown(alphas, betas, i + 1)
            if cut is not None:
                t = cut
                break
    if t == steps and steps > 1:
        cut = _first_breakdown(alphas, betas, steps - 1)
        if cut is not None:
            t = cut
    return Qb[:t].t(), _tridiagonal(alphas, betas, t)


class PredictionCache:
    """What does not depend on the test points, computed once: the mean cache alpha = (s K + sigma^2 I)^-1 (y - mu)
    (one CG solve, preconditioned at rank `pre_size`) and the variance cache Q, chol(T) of `lanc_iter` Lanczos steps
    started at y - mu.  GPyTorch keeps the same two caches on the model between calls in eval mode (its prediction
    strategy: the reference's test() on the validation split and then on the test split, train_simplexgp.py:123-165,
    solves and runs Lanczos once per epoch, not once per split); `fit` builds one cache per evaluated epoch.
    The cache holds for the hyper-parameters, x and y it was built from: build a new one after they change."""

    @torch.no_grad()
    def __init__(self, model, x, y, max_cg_iter=1000, cg_tol=1e-2, lanc_iter=100, variance=True, pre_size=100):
        assert isinstance(model, LatticeGP)
        self.model, self.x = model, x
        r = (y - model.mean).reshape(-1, 1)
        K = model.kernel(x, x)                            # one operator (one lattice) for the preconditioner and the solve
        precond = model.preconditioner(x, pre_size, K=K) if pre_size > 0 else None
        self.alpha, self.solve_info = model.khat_solve(x, r, K=K, max_iter=max_cg_iter, tol=cg_tol, precond=precond)
        self.Q = self.chol = None
        if variance:
            # Lanczos on the rows in the lattice's own order where the HIP path offers it (dot products do not care about
            # the row order; every MVM skips its two row permutations), the basis permuted back once
            with model.khat_in_lattice_rows(x, K=K) as (mm_rows, to_rows, from_rows):
                Q, T = lanczos(mm_rows, to_rows(r).squeeze(-1), min(lanc_iter, x.shape[0]))
                self.Q = from_rows(Q.contiguous())
            jitter = 1e-6 * T.diagonal().abs().max()
            self.chol = torch.linalg.cholesky(T + jitter * torch.eye(T.shape[0], dtype=T.dtype, device=T.device))

    @torch.no_grad()
    def predict(self, x_star):
        model = self.model
        K_star = model.kernel(x_star, self.x)             # RectangularLazyLattice, [n*, n]
        s = model.outputscale
        if self.Q is None:
            return model.mean + s * K_star.matmul(self.alpha).squeeze(-1), None
        # the mean column and the t variance columns through the rectangular operator in ONE filter
        KAQ = s * K_star.matmul(torch.cat([self.alpha, self.Q], 1).contiguous())          # [n*, 1 + t]
        mean = model.mean + KAQ[:, 0]
        proj = torch.linalg.solve_triangular(self.chol, KAQ[:, 1:].t(), upper=False)
        prior = s * model.kernel(x_star, x_star, diag=True)
        return mean, (prior - (proj ** 2).sum(0)).clamp_min(1e-8)


@torch.no_grad()
def predict(model, x, y, x_star, max_cg_iter=1000, cg_tol=1e-2, lanc_iter=100, variance=True, pre_size=100, cache=None):
    """Posterior mean and variance of the latent function at x_star.

    mean = mu + s K(x*, X) (s K + sigma^2 I)^-1 (y - mu)          one CG solve + one rectangular MVM
    var  = s k(x*, x*) - || L^-1 Q^T (s K(X, x*)) ||^2            K^-1 ~ Q T^-1 Q^T from `lanc_iter`
           Lanczos steps started at y - mu, T = L L^T; one rectangular MVM with lanc_iter columns
    pre_size: rank of the pivoted-Cholesky preconditioner of the mean solve (the reference's test() default,
    train_simplexgp.py:60: 100; 0 = plain CG)
    cache: a PredictionCache of (model, x, y) to predict from (the solve and the Lanczos run are then not repeated)
    """
    if cache is None:
        cache = PredictionCache(model, x, y, max_cg_iter=max_cg_iter, cg_tol=cg_tol, lanc_iter=lanc_iter, variance=variance,
                                pre_size=pre_size)
    return cache.predict(x_star)


@torch.no_grad()
def evaluate(model, x, y, x_star, y_star, label="test", **predict_args):
    """RMSE, MAE and mean negative log predictive density (train_simplexgp.py:74-84)."""
    mean, var = predict(model, x, y, x_star, **predict_args)
    rmse = (mean - y_star).pow(2).mean(0).sqrt()
    mae = (mean - y_star).abs().mean(0)
    nll = -torch.distributions.Normal(mean, (var + model.noise).sqrt()).log_prob(y_star).mean()
    rmse, mae, nll = torch.stack([rmse, mae, nll]).tolist()                  # one device read, not three
    return {f"{label}/rmse": rmse, f"{label}/mae": mae, f"{label}/nll": nll}


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
            # one mean / variance cache for both splits (what GPyTorch's eval mode keeps between the reference's two test() calls)
            cache = PredictionCache(model, x, y, max_cg_iter=cg_iter, cg_tol=cg_eval_tol, lanc_iter=lanc_iter, pre_size=pre_size)
            row.update(evaluate(model, x, y, val[0], val[1], label="val", cache=cache))
            if test is not None:
                row.update(evaluate(model, x, y, test[0], test[1], label="test", cache=cache))
            del cache
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
