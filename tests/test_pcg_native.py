"""The preconditioned solve of the reference's training recipe (experiments/train_simplexgp.py:34-41:
max_preconditioner_size(100)) on the native path: plx_pchol_* (batched pivoted Cholesky), plx_pcg_* (the
preconditioner's two passes and the direction update) against the torch formulation of the same algorithm
(solvers.PivotedCholeskyPreconditioner / _batched_pcg, which run on CPU tensors too and are what
tests/test_solvers.py checks against dense linear algebra), and at config-3 size against the CPU oracle's MVM."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def plx():
    import simplex_gp_amd as plx
    assert torch.cuda.is_available()
    return plx


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("n,kp,k,t", [(1000, 32, 20, 1), (4097, 112, 100, 12), (777, 16, 16, 5), (20000, 144, 130, 16),
                                      (64, 16, 3, 4)])
def test_pcg_passes_against_torch(plx, n, kp, k, t, half):
    """plx_pcg_project (MFMA gram + fp64 solve), plx_pcg_apply (+ <R, Z>) and plx_pcg_step_direction on a random
    factor: against the same expressions in fp64."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    g = torch.Generator().manual_seed(n + t)
    ld = (n + 63) // 64 * 64
    Lt = torch.zeros(kp, ld)
    Lt[:k, :n] = torch.randn(k, n, generator=g) * 0.3
    if half:
        Lt = Lt.half().float()             # the factor the fp16 path stores; the expressions below use the same values
    R = torch.randn(n, t, generator=g)
    noise = 0.37
    C = (Lt[:k, :n].double() @ Lt[:k, :n].double().T) + noise * torch.eye(k, dtype=torch.float64)
    cinv = torch.eye(kp, dtype=torch.float64) / noise
    cinv[:k, :k] = torch.linalg.inv(C)
    Lt_d, R_d, cinv_d = Lt.cuda(), R.cuda(), cinv.cuda()
    T = torch.zeros(kp, 16, device="cuda")
    Z = torch.empty_like(R_d)
    rz = torch.empty(t, device="cuda")
    work = torch.empty(int(lib.plx_pcg_work_floats(n, kp, t)), device="cuda")
    scale = torch.tensor([1.0, 1.0 / noise], device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ftype = nv.FACTOR_F32
    if half:
        Lh = torch.empty(kp, ld, dtype=torch.float16, device="cuda")
        nv.check(lib.plx_pcg_factor_to_half(_vp(Lt_d), ld, kp, _vp(Lh), stream), "to_half")
        assert torch.equal(Lh.float(), Lt_d)
        Lt_d, ftype = Lh, nv.FACTOR_F16
    nv.check(lib.plx_pcg_project(_vp(Lt_d), ftype, ld, kp, _vp(R_d), n, t, _vp(cinv_d), _vp(T), _vp(work), stream), "project")
    nv.check(lib.plx_pcg_apply(_vp(Lt_d), ftype, ld, kp, k, _vp(R_d), n, t, _vp(T), _vp(scale), _vp(Z), _vp(rz), _vp(work), stream),
             "apply")
    G = Lt[:k, :n].double() @ R.double()
    T_want = torch.linalg.solve(C, G)
    Z_want = (R.double() - Lt[:k, :n].double().T @ T_want) / noise
    assert torch.allclose(T[:k, :t].cpu().double(), T_want, rtol=2e-4, atol=2e-5 * float(T_want.abs().max()))
    err = float((Z.cpu().double() - Z_want).norm() / Z_want.norm())
    assert err < 2e-5, err
    rz_want = (R.double() * Z_want).sum(0)
    assert torch.allclose(rz.cpu().double(), rz_want, rtol=2e-4)
    # P^-1 really is the inverse of L L^T + noise I
    back = Lt[:k, :n].double().T @ (Lt[:k, :n].double() @ Z.cpu().double()) + noise * Z.cpu().double()
    assert float((back - R.double()).norm() / R.double().norm()) < 5e-4          # cond(P) ~ 5e3 here: fp32 rounding of Z, amplified
    # direction: beta = rz' / rz on active columns, P = Z + beta P, activity from the true residual
    P = torch.randn(n, t, generator=g).cuda()
    P0 = P.clone()
    rz_old = (torch.rand(t, generator=g) + 0.5).cuda()
    rr = (torch.rand(t, generator=g) * 2).cuda()
    b_norm = torch.ones(t).cuda()
    active = (torch.arange(t) % 3 != 1).float().cuda()
    beta, active_out = torch.empty(t, device="cuda"), torch.empty(t, device="cuda")
    nv.check(lib.plx_pcg_step_direction(_vp(P), _vp(Z), _vp(rz), _vp(rz_old), _vp(rr), _vp(active), _vp(b_norm), 1.0, n, t,
                                        _vp(beta), _vp(active_out), stream), "direction")
    beta_want = torch.where(active > 0, rz / rz_old, torch.zeros_like(rz))
    assert torch.allclose(beta, beta_want, rtol=1e-6)
    assert torch.allclose(P, Z + P0 * beta_want, rtol=1e-6, atol=1e-6)
    assert torch.equal(active_out, ((active > 0) & (rr.sqrt() > 1.0)).float())


@pytest.mark.parametrize("n,d,ell,rank,noise_raw", [(20000, 3, 0.5, 40, 0.0), (300, 1, 0.6, 30, -3.0), (5000, 8, 0.8, 33, 0.0),
                                                    (2500, 2, 1.0, 100, -2.0), (50, 2, 1.0, 100, 0.0)])
def test_batched_factor_is_the_sequential_pivoted_cholesky(plx, n, d, ell, rank, noise_raw):
    """LatticePreconditioner (speculated pivot batches, lattice row order, native passes) against the sequential torch
    algorithm through the same HIP operator (one single-column MVM per pivot, caller row order): the same factor column
    by column, the same log-determinant, the same solve and the same samples."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(7 * n + d)
    x = (torch.randn(n, d, generator=g) if d > 1 else torch.rand(n, 1, generator=g) * 6).cuda()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        model.kernel.lengthscale = ell
        model.raw_noise.fill_(noise_raw)
        K = model.kernel(x, x)
        want = solvers.PivotedCholeskyPreconditioner(K.matmul, n, model.outputscale, model.noise, rank, device=x.device,
                                                     dtype=x.dtype)
        got = model.preconditioner(x, rank, K=K, factor_dtype=torch.float32)
        got16 = model.preconditioner(x, rank, K=K)
    assert isinstance(got, solvers.LatticePreconditioner) and got16.factor_type == 1 and got.factor_type == 0
    k = min(rank, n)
    assert got.rank == want.rank == k
    print(f"n={n} d={d} rank={k}: {got.batches} batches")
    Lg, Lw = got.L, want.L
    scale = float(Lw.abs().max())
    assert float((Lg - Lw).abs().max()) <= 2e-4 * scale, float((Lg - Lw).abs().max()) / scale
    assert abs(float(got.logdet()) - float(want.logdet())) <= 1e-5 * n + 1e-3
    R = torch.randn(n, 3, generator=g).cuda()
    a, b = got.solve(R), want.solve(R)
    assert float((a - b).norm() / b.norm()) < 5e-4
    ga, gb = torch.Generator(device="cuda").manual_seed(3), torch.Generator(device="cuda").manual_seed(3)
    sa, sb = got.sample(4, generator=ga), want.sample(4, generator=gb)
    assert float((sa - sb).norm() / sb.norm()) < 1e-4
    if n >= 5000:
        assert got.batches < k          # the speculation pays: several pivots per MVM
    # the default keeps the factor in fp16: the same factor rounded once, and a preconditioner that is consistent with
    # ITS factor -- solve() inverts L~ L~^T + sigma^2 I for the stored L~, logdet() is that matrix's
    L16 = got16.L
    assert float((L16 - Lg).abs().max()) <= 1e-3 * scale
    assert torch.equal(L16, Lg.half().float())
    noise = float(model.noise)
    a16 = got16.solve(R)
    back = L16.double() @ (L16.double().T @ a16.double()) + noise * a16.double()
    assert float((back - R.double()).norm() / R.double().norm()) < 2e-3
    Cd = L16.double().T @ L16.double() + noise * torch.eye(k, dtype=torch.float64, device="cuda")
    assert abs(float(got16.logdet()) - float(torch.logdet(Cd) + (n - k) * math.log(noise))) <= 1e-6 * n + 1e-4
    plx.lattice_cache().clear()


def test_native_pcg_matches_torch_pcg(plx):
    """The preconditioned solve on the native path (lattice row order, plx_pcg_* passes, device-side coefficients)
    against the torch formulation in caller row order over the same operator: same iteration count within one check
    interval, same solution, same Lanczos tridiagonals."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(11)
    n, d = 30000, 2
    x = torch.randn(n, d, generator=g).cuda()
    y = torch.sin(2 * x[:, 0]) + 0.05 * torch.randn(n, generator=g).cuda()
    Zp = (torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1).cuda()
    rhs = torch.cat([y[:, None], Zp], 1).contiguous()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        model.raw_noise.fill_(-4.0)
        K = model.kernel(x, x)
        pre_t = solvers.PivotedCholeskyPreconditioner(K.matmul, n, model.outputscale, model.noise, 50, device=x.device,
                                                      dtype=x.dtype)
        pre_n = model.preconditioner(x, 50, K=K, factor_dtype=torch.float32)
        pre_h = model.preconditioner(x, 50, K=K)                 # the default: the factor kept in fp16
        sol_t, info_t = model.khat_solve(x, rhs, K=K, max_iter=60, tol=0.0, precond=pre_t, want_tridiag=True)
        sol_n, info_n = model.khat_solve(x, rhs, K=K, max_iter=60, tol=0.0, precond=pre_n, want_tridiag=True)
        plain, info_p = model.khat_solve(x, rhs, K=K, max_iter=60, tol=0.0)
    assert info_t["iterations"] == info_n["iterations"] == 60
    assert float((sol_t - sol_n).norm() / sol_t.norm()) < 1e-3
    assert torch.allclose(info_t["rz0"], info_n["rz0"], rtol=1e-3)
    # the first Lanczos coefficients agree closely (later ones drift apart at fp32 rounding, like any two CG runs)
    Tt, Tn = info_t["tridiag"][:, :8, :8], info_n["tridiag"][:, :8, :8]
    assert float((Tt - Tn).abs().max() / Tt.abs().max()) < 1e-2
    # and the preconditioner does its job at this noise level: a smaller residual than plain CG after 60 iterations
    assert float(info_n["residual"][0]) < float(info_p["residual"][0])
    # single right-hand side (prediction: training.predict solves for one column)
    with torch.no_grad():
        s1, i1 = model.khat_solve(x, rhs[:, :1].contiguous(), K=K, max_iter=40, tol=0.0, precond=pre_n)
        s1t, _ = model.khat_solve(x, rhs[:, :1].contiguous(), K=K, max_iter=40, tol=0.0, precond=pre_t)
        sol_h, info_h = model.khat_solve(x, rhs, K=K, max_iter=60, tol=0.0, precond=pre_h, want_tridiag=True)
    assert float((s1 - s1t).norm() / s1t.norm()) < 1e-3
    # fp16 factor: another (equally valid) preconditioner, 5e-4 away from the fp32 one.  What has to hold is that the
    # iteration is consistent -- the residual it reports is the residual of its solution -- and converges alike (single
    # columns of two CG runs on this ill-conditioned system differ by 2x at any given iteration; their mean does not)
    assert pre_h.factor_type == 1
    with torch.no_grad():
        true_h = (model.khat_matmul(x, K)(sol_h) - rhs).norm(dim=0) / rhs.norm(dim=0)
    assert float((true_h - info_h["residual"]).abs().max()) < 2e-3      # fp32 recurrence drift over 60 ill-conditioned iterations
    ratio = float((info_h["residual"].log() - info_n["residual"].log()).mean().exp())
    assert 0.6 < ratio < 1.6, ratio
    # round 6: the iteration without its three stand-alone reductions (pAp, |R|^2, <R, Z> added up inside the update and the
    # direction kernels) against the one with them: the same iteration to rounding, reproducible bit for bit
    default = solvers.FUSED_CG_STEPS
    assert default == "auto" and solvers._fuse_cg_steps(n)                 # (30,000 rows: the fold is on by itself)
    with torch.no_grad():
        try:
            solvers.FUSED_CG_STEPS = True
            sol_f2, _ = model.khat_solve(x, rhs, K=K, max_iter=60, tol=0.0, precond=pre_n, want_tridiag=True)
            solvers.FUSED_CG_STEPS = False
            sol_u, info_u = model.khat_solve(x, rhs, K=K, max_iter=60, tol=0.0, precond=pre_n, want_tridiag=True)
        finally:
            solvers.FUSED_CG_STEPS = default
    assert torch.equal(sol_f2, sol_n)
    assert float((sol_u - sol_n).norm() / sol_n.norm()) < 1e-3
    assert torch.allclose(info_u["rz0"], info_n["rz0"], rtol=1e-5)
    Tu = info_u["tridiag"][:, :8, :8]
    assert float((Tu - Tn).abs().max() / Tn.abs().max()) < 1e-2
    true_n = (model.khat_matmul(x, K)(sol_n) - rhs).norm(dim=0) / rhs.norm(dim=0)
    assert float((true_n - info_n["residual"]).abs().max()) < 2e-3        # rr as the fused direction kernel stores it
    plx.lattice_cache().clear()


def test_config3_full_size_preconditioned_cg(plx):
    """BASELINE.json configs[2] with the reference's training settings (train_simplexgp.py:34-41: pre_size 100,
    [y | 10 probes] drawn from N(0, P)): N = 1e6, d = 8, 20 preconditioned CG iterations on one lattice build; the
    residual the native solve reports is reproduced by (s K + sigma^2 I) x recomputed with the CPU oracle's filter."""
    from simplex_gp_amd import solvers
    n, d, iters = 1_000_000, 8, 20
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g)
    y = torch.randn(n, generator=g)
    xc = x.cuda()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    cache = plx.lattice_cache()
    cache.clear()
    with torch.no_grad():
        K = model.kernel(xc, xc)
        pre = model.preconditioner(xc, 100, K=K)
        assert isinstance(pre, solvers.LatticePreconditioner) and pre.rank == 100
        print("config 3 preconditioner: 100 pivots in", pre.batches, "batches")
        assert pre.batches <= 20
        Zp = pre.sample(10, generator=torch.Generator(device="cuda").manual_seed(0))
        rhs = torch.cat([y.cuda()[:, None], Zp], 1).contiguous()
        sol, info = model.khat_solve(xc, rhs, K=K, max_iter=iters, tol=0.0, precond=pre, want_tridiag=True)
        plain, info_p = model.khat_solve(xc, rhs, K=K, max_iter=iters, tol=0.0)
    assert info["iterations"] == iters and info["tridiag"].shape == (11, iters, iters)
    reported = info["residual"].cpu().numpy()
    s, noise = float(model.outputscale), float(model.noise)
    ell = model.kernel.lengthscale.detach().cpu()
    ref = (x / ell).contiguous().numpy()
    taps = model.kernel.dkernel_fn.get_coeffs().numpy()
    sol_cpu, rhs_cpu = sol.cpu().numpy(), rhs.cpu().numpy()
    oracle.set_exact_mode(False)
    try:
        Kx = oracle.filter(sol_cpu, ref, taps)
    finally:
        oracle.set_exact_mode(True)
    # fp64 throughout: numpy reduces a C-ordered fp32 [n, 11] array along axis 0 row by row (1e6 sequential fp32 adds)
    resid = rhs_cpu.astype(np.float64) - (s * Kx.astype(np.float64) + noise * sol_cpu.astype(np.float64))
    true_rel = np.linalg.norm(resid, axis=0) / np.linalg.norm(rhs_cpu.astype(np.float64), axis=0)
    print("config 3 (pre_size 100): native residual", reported, "oracle-recomputed", true_rel, "plain CG",
          info_p["residual"].cpu().numpy())
    assert np.abs(true_rel - reported).max() <= 1e-4
    assert reported.max() < 0.5          # 20 iterations at cg_tol = 1 (the reference's training setting): a loose solve by design
    cache.clear()


def test_onehot_splat_and_recycled_lattice_guard(plx):
    """plx_splat_onehot (the factor build's kernel rows: d + 1 numbers per column instead of a pass over all corners)
    equals the general splat of the same one-hot right-hand side; and a LatticePreconditioner whose lattice object has
    been rebuilt for other positions (lattice-cache eviction) refuses to be applied instead of permuting rows wrongly."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(9)
    n, d = 20000, 4
    x = torch.randn(n, d, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice().build(x, taps)
    lat.set_lattice_row_order(True)
    for t, nb in ((1, 1), (4, 3), (12, 12), (16, 13)):
        pts = torch.randperm(n, generator=g)[:nb].to(torch.int32).cuda()
        rhs = torch.zeros(n, t, device="cuda")
        rhs[pts.long(), torch.arange(nb, device="cuda")] = 1.0
        want = lat.splat(rhs)
        got = lat.splat_onehot(pts, nb, lat.new_values(t), vd=t)
        assert torch.equal(got, want), (t, nb)
        # plx_pchol_onehot writes the same right-hand side (for callers that only have plx_apply)
        from simplex_gp_amd import _native as nv
        rhs2 = torch.full((n, t), 7.0, device="cuda")
        nv.check(nv.lib().plx_pchol_onehot(_vp(pts), nb, n, t, _vp(rhs2), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                 "plx_pchol_onehot")
        assert torch.equal(rhs2, rhs)
    lat.set_lattice_row_order(False)
    lat.close()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        pre = model.preconditioner(x, 20)
        R = torch.randn(n, 2, generator=g).cuda()
        z = pre.solve(R)
        assert torch.isfinite(z).all()
        pre.lat.build((x * 1.7).contiguous(), taps)                      # what an eviction from the lattice cache does to the object
        with pytest.raises(RuntimeError, match="rebuilt"):
            pre.solve(R)
    plx.lattice_cache().clear()


@pytest.mark.parametrize("n,d,ell,order", [(20000, 4, 1.0, 1), (30000, 8, 0.5, 1), (3000, 3, 0.3, 2), (5000, 12, 0.8, 1),
                                           (2000, 2, 0.05, 3), (40000, 6, 2.5, 1), (1500, 18, 1.0, 1)])
def test_filter_onehot_on_the_frontier_equals_the_dense_stages(plx, n, d, ell, order):
    """plx_filter_onehot (kernel rows K e_p on the frontier of their non-zero vertex rows) against the dense
    splat_onehot + blur + slice it replaces (same operations in the same order: equal bits for multi-column rows) and
    against plx_apply of the one-hot right-hand side; both row orders; candidates that sit next to each other (shared
    vertices between columns), repeated calls on the same lattice (the position map is reset per call), and a lattice
    coarse enough that the frontier ends up holding every vertex."""
    g = torch.Generator().manual_seed(n + d)
    x = (torch.randn(n, d, generator=g) / ell).cuda()
    half = np.linspace(0.2, 0.7, order).astype(np.float32)
    taps = np.concatenate([half, [1.0], half[::-1]]).astype(np.float32)
    lat = plx.Lattice().build(x, taps)
    from simplex_gp_amd import _native as nv
    perm = torch.from_numpy(lat.export(nv.ARRAY_POINT_PERM).astype(np.int64)).cuda()
    frontier = torch.zeros(1, dtype=torch.int32, device="cuda")
    for lattice_rows in (True, False):
        lat.set_lattice_row_order(lattice_rows)
        for t, nb in ((1, 1), (4, 3), (12, 12), (16, 13), (8, 8)):
            pts = torch.randperm(n, generator=g)[:nb].to(torch.int32)
            if nb >= 3:
                pts[1] = (pts[0] + 1) % n          # neighbours in lattice order: their simplices share vertices
                pts[2] = pts[0]                    # ... and the same point twice
            pts = pts.cuda()
            vals, scratch = lat.new_values(t), lat.new_values(t)
            dense = torch.full((n, t), 3.0, device="cuda")
            lat.filter_onehot(pts, nb, vals, scratch, dense, vd=t, sparse=False, frontier=frontier)
            assert int(frontier.item()) == lat.m
            got = torch.full((n, t), 5.0, device="cuda")
            lat.filter_onehot(pts, nb, vals, scratch, got, vd=t, sparse=True, frontier=frontier)
            f = int(frontier.item())
            assert (d + 1) <= f <= lat.m
            rhs = torch.zeros(n, t, device="cuda")
            rows = pts.long() if lattice_rows else perm[pts.long()]
            rhs[rows, torch.arange(nb, device="cuda")] = 1.0
            want = lat.apply(rhs)
            if t > 1:
                assert torch.equal(got, dense), (lattice_rows, t, nb, float((got - dense).abs().max()))
            scale = float(want.abs().max())
            assert float((got - dense).abs().max()) <= 2e-6 * scale
            assert float((got - want).abs().max()) <= 2e-6 * scale, (lattice_rows, t, nb)
            assert got[:, nb:].abs().max() == 0 if nb < t else True
    lat.set_lattice_row_order(False)
    lat.close()


@pytest.mark.parametrize("n,d,ell,rank", [(20000, 3, 0.5, 40), (50000, 6, 0.4, 60), (3000, 2, 2.0, 30)])
def test_factor_batch_modes_build_the_same_factor(plx, n, d, ell, rank):
    """The factor does not depend on HOW its batches are run: planned steps only, planned steps + one launch per
    remaining candidate, the adaptive default; kernel rows on the frontier or by the dense stages; batches of 1
    (the sequential algorithm itself), 5 or 16 instead of 12 -- same pivots, entries within rounding."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(n + rank)
    x = torch.randn(n, d, generator=g).cuda()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        model.kernel.lengthscale = ell
        ref = model.preconditioner(x, rank, factor_dtype=torch.float32)
        lat, s, noise = ref.lat, float(model.outputscale), float(model.noise)
        want = ref.L
        seen = []
        for kw in (dict(batch=1), dict(exact_steps=True), dict(exact_steps=False), dict(sparse_rows=False),
                   dict(sparse_rows=False, exact_steps=True, batch=16), dict(batch=5, exact_steps=False)):
            got = solvers.LatticePreconditioner(lat, s, noise, rank, factor_dtype=torch.float32, **kw)
            # (batches that end elsewhere split a column's sum differently between the panel update and the in-batch
            # corrections: same pivots, entries within rounding)
            assert float((got.L - want).abs().max()) <= 2e-6 * float(want.abs().max()), kw
            if kw == dict(sparse_rows=False) and got.batches == ref.batches:
                assert torch.equal(got.L, want)       # same batches, kernel rows by the dense stages: equal bits
            seen.append((kw, got.batches, got.planned_batches))
        print(seen)
        one = seen[0]
        assert one[1] == min(rank, n)                 # batch = 1: one pivot per batch
    plx.lattice_cache().clear()


def test_filter_onehot_argument_checks_and_replay_fallback(plx):
    """plx_filter_onehot refuses what it cannot do (aliased work buffers, more than 16 columns, NULL output) and, on a
    lattice built under plx_tune("reference_growth", 1) -- whose patched neighbour table is not symmetric --, runs the
    dense stages whatever `sparse` says (frontier = m) with the result of plx_apply."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    g = torch.Generator().manual_seed(5)
    n, d = 6000, 5
    x = (torch.randn(n, d, generator=g) * 2).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lat = plx.Lattice().build(x, taps)
    pts = torch.arange(4, dtype=torch.int32, device="cuda")
    vals, scratch = lat.new_values(4), lat.new_values(4)
    out = torch.empty(n, 4, device="cuda")
    # (PLX_ERR_INVALID = 1)
    assert lib.plx_filter_onehot(lat._h, _vp(pts), 4, 4, _vp(vals), _vp(vals), _vp(out), 1, None, stream) == 1
    assert lib.plx_filter_onehot(lat._h, _vp(pts), 4, 4, _vp(vals), _vp(scratch), None, 1, None, stream) == 1
    assert lib.plx_filter_onehot(lat._h, _vp(pts), 5, 4, _vp(vals), _vp(scratch), _vp(out), 1, None, stream) == 1
    big_v, big_s = lat.new_values(20), lat.new_values(20)
    big_o = torch.empty(n, 20, device="cuda")
    many = torch.arange(17, dtype=torch.int32, device="cuda")
    assert lib.plx_filter_onehot(lat._h, _vp(many), 17, 20, _vp(big_v), _vp(big_s), _vp(big_o), 1, None, stream) == 1
    # vd = 3 (neither 1 nor a multiple of 4): the dense stages, silently
    v3, s3, o3 = lat.new_values(3), lat.new_values(3), torch.empty(n, 3, device="cuda")
    frontier = torch.zeros(1, dtype=torch.int32, device="cuda")
    lat.filter_onehot(pts, 3, v3, s3, o3, vd=3, sparse=True, frontier=frontier)
    assert int(frontier.item()) == lat.m
    lat.close()
    nv.check(lib.plx_tune(b"reference_growth", 1), "plx_tune")
    try:
        # a cloud whose reference table grows (the quirk needs > 2^14 vertices)
        xg = (torch.randn(4000, 8, generator=g) * 3).cuda()
        lat = plx.Lattice().build(xg, taps)
        info = lat.reference_growth_info()
        assert info["replayed"]
        vals, scratch = lat.new_values(4), lat.new_values(4)
        got = torch.empty(4000, 4, device="cuda")
        lat.filter_onehot(pts, 4, vals, scratch, got, vd=4, sparse=True, frontier=frontier)
        assert int(frontier.item()) == lat.m
        rhs = torch.zeros(4000, 4, device="cuda")
        perm = torch.from_numpy(lat.export(nv.ARRAY_POINT_PERM).astype(np.int64)).cuda()
        rhs[perm[pts.long()], torch.arange(4, device="cuda")] = 1.0
        want = lat.apply(rhs)
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
        lat.close()
    finally:
        nv.check(lib.plx_tune(b"reference_growth", 0), "plx_tune")
