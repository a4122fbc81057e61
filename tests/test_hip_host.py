"""Host mirror + sharded structure on a real MI355X, all through libplx."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def plx():
    import simplex_gp_amd as plx
    assert torch.cuda.is_available()
    return plx


CASES = {
    "n50_d3_L2_rbf_o1": ("rbf", 1), "n200_d1_L1_rbf_o1": ("rbf", 1),
    "n50_d3_L2_matern15_o3": ("matern15", 3), "n200_d1_L1_matern15_o3": ("matern15", 3),
    "n300_d4_L3_rbf_o2": ("rbf", 2),
}


@pytest.mark.parametrize("cname", sorted(CASES))
def test_autograd_on_gpu_matches_reference(plx, golden_dir, cname):
    """LatticeFilterGeneral forward + both gradients (py:76-124) through the HIP filter."""
    host = np.load(os.path.join(golden_dir, "host_side.npz"))
    profiles = {"rbf": plx.rbf, "matern15": lambda d2: plx.Matern.apply(d2, 1.5)}
    pname, order = CASES[cname]
    dk = plx.DiscretizedKernelFN(profiles[pname], order)
    x = torch.from_numpy(host[f"autograd/{cname}/x"]).cuda().requires_grad_(True)
    s = torch.from_numpy(host[f"autograd/{cname}/src"]).cuda().requires_grad_(True)
    gout = torch.from_numpy(host[f"autograd/{cname}/grad_out"]).cuda()
    assert plx.LatticeFilterGeneral.method is None          # the HIP path, not a test hook
    out = plx.LatticeFilterGeneral.apply(s, x, dk)
    out.backward(gout)
    for got, name in [(out, "out"), (s.grad, "grad_src"), (x.grad, "grad_x")]:
        assert rel_l2(got.detach().cpu().numpy(), host[f"autograd/{cname}/{name}"]) <= 2e-5, name
    s2 = s.detach().clone().requires_grad_(True)
    plx.LatticeFilterGeneral.apply(s2, x.detach(), dk).backward(gout)
    assert rel_l2(s2.grad.cpu().numpy(), host[f"autograd/{cname}/grad_src_only"]) <= 2e-5


def test_kernel_matmul_and_lattice_reuse(plx):
    """K(x, x) @ V through RBFLattice; repeated MVMs on the same x reuse one lattice."""
    torch.manual_seed(0)
    k = plx.RBFLattice(order=1, ard_num_dims=3).cuda()
    x = torch.randn(5000, 3, device="cuda")
    V = torch.randn(5000, 4, device="cuda")
    cache = plx.lattice_cache()
    cache.clear()
    with torch.no_grad():
        K = k(x, x)
        h0, m0 = cache.hits, cache.misses
        outs = [K.matmul(V) for _ in range(5)]
    assert cache.misses == m0 + 1 and cache.hits == h0 + 4            # one build, four reuses
    assert all(torch.equal(o, outs[0]) for o in outs)
    want = oracle.filter(V.cpu().numpy(), (x / k.lengthscale).detach().cpu().numpy(),
                         k.dkernel_fn.get_coeffs().numpy())
    assert rel_l2(outs[0].cpu().numpy(), want) <= 1e-5
    # in-place change of the positions bumps tensor._version -> new lattice, not a stale hit
    xs = (x / k.lengthscale).detach().clone()
    lat_out = plx.lattice_kernel.cached_filter(V, xs, k.dkernel_fn.get_coeffs())
    xs.mul_(0.5)
    lat_out2 = plx.lattice_kernel.cached_filter(V, xs, k.dkernel_fn.get_coeffs())
    want2 = oracle.filter(V.cpu().numpy(), xs.cpu().numpy(), k.dkernel_fn.get_coeffs().numpy())
    assert rel_l2(lat_out2.cpu().numpy(), want2) <= 1e-5 and not torch.equal(lat_out, lat_out2)
    # rectangular (prediction) path
    xs_test = torch.randn(700, 3, device="cuda")
    with torch.no_grad():
        R = k(xs_test, x)
        m1 = cache.misses
        got = R.matmul(V)
        again = R.matmul(V)
    assert cache.misses == m1 + 1 and torch.equal(got, again)         # the stacked points are made once per operator: one build
    ell = k.lengthscale.detach()
    big_x = torch.cat([x / ell, xs_test / ell]).cpu().numpy()
    big_v = np.concatenate([V.cpu().numpy(), np.zeros((700, 4), np.float32)])
    assert rel_l2(got.cpu().numpy(), oracle.filter(big_v, big_x, k.dkernel_fn.get_coeffs().numpy())[5000:]) <= 1e-5


@pytest.mark.parametrize("shards,vd", [(2, 3), (3, 3), (2, 7), (3, 20)])
def test_owned_ranges_compose(plx, shards, vd):
    """plx_build with an owned row range (the multi-GPU structure), emulated on one GPU:
    per-shard splats add up to the full splat, per-shard slices tile the full output."""
    from simplex_gp_amd.distributed import shard_bounds
    g = torch.Generator().manual_seed(3)
    n, d = 20011, 4          # vd 3: scan splat, 7 and 20: lane-group splat, on partially covered vertex sets
    x = torch.randn(n, d, generator=g).cuda()
    v = torch.randn(n, vd, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    full = plx.Lattice().build(x, taps)
    want = full.apply(v)
    total = None
    lats = []
    for r in range(shards):
        lo, hi = shard_bounds(n, shards, r)
        lat = plx.Lattice().build(x, taps, shard=(r, shards))
        assert lat.m == full.m and lat.n_owned == hi - lo
        part = lat.splat(v[lo:hi])
        total = part.clone() if total is None else total + part
        lats.append((lat, lo, hi))
    # NB: a sharded build orders points shard by shard, so its vertex numbering differs from the
    # single-shard lattice `full`; all shards of ONE job agree with each other (checked via keys).
    from simplex_gp_amd import _native as nv
    keys0 = lats[0][0].export(nv.ARRAY_KEYS)
    assert all(np.array_equal(keys0, lat.export(nv.ARRAY_KEYS)) for lat, _, _ in lats[1:])
    got = torch.empty_like(want)
    for lat, lo, hi in lats:
        blurred = lat.blur(total.clone(), vd=vd)
        got[lo:hi] = lat.slice(blurred, vd=vd)
    assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) <= 1e-6
    oracle.set_exact_mode(False)
    ref = oracle.filter(v.cpu().numpy(), x.cpu().numpy(), taps)
    oracle.set_exact_mode(True)
    assert rel_l2(got.cpu().numpy(), ref) <= 1e-5
    # an empty shard is legal (more shards than rows)
    tiny = plx.Lattice().build(x[:2].contiguous(), taps, shard=(2, 3))
    assert tiny.n_owned == 0 and float(tiny.splat(v[:0]).abs().sum()) == 0.0


def test_sharded_mvm_single_process(plx):
    """ShardedLatticeMVM without a process group degenerates to the plain MVM."""
    from simplex_gp_amd.distributed import ShardedLatticeMVM
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3000, 3, generator=g).cuda()
    v = torch.randn(3000, 2, generator=g).cuda()
    taps = np.array([0.5, 1.0, 0.5], np.float32)
    op = ShardedLatticeMVM(x, taps)
    assert torch.equal(op.matmul(v), plx.Lattice().build(x, taps).apply(v))


@pytest.mark.parametrize("pre_size", [0, 100])
def test_snelson_config1_on_gpu(plx, golden_dir, pre_size):
    """BASELINE.json config 1 on the HIP path: |MLL_lattice - MLL_exact| < 0.1 (tests/train_snelson.py:96).  pre_size = 100 is
    what the reference's own test trains with (max_preconditioner_size(100), tests/train_snelson.py:48-55): the native
    pivoted-Cholesky preconditioner then runs inside every training solve (rank 100 of n = 200)."""
    from simplex_gp_amd import solvers
    sn = np.loadtxt(os.path.join(golden_dir, "snelson.csv"), delimiter=",", skiprows=1).astype(np.float32)
    x, y = torch.from_numpy(sn[:, :1].copy()).cuda(), torch.from_numpy(sn[:, 1].copy()).cuda()
    torch.manual_seed(0)
    exact = solvers.ExactRBFGP().cuda()
    opt = torch.optim.Adam(exact.parameters(), lr=0.1)
    for _ in range(100):
        opt.zero_grad()
        (-exact.mll(x, y)).backward()
        opt.step()
    exact_mll = float(exact.mll(x, y))
    model = solvers.LatticeGP(plx.RBFLattice(order=1)).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=0.1)
    for i in range(100):
        opt.zero_grad()
        (-solvers.marginal_log_likelihood(model, x, y, num_probes=10, cg_tol=1e-4, max_cg_iter=500, seed=i,
                                          pre_size=pre_size)).backward()
        opt.step()
    with torch.no_grad():
        lattice_mll = float(solvers.marginal_log_likelihood(model, x, y, num_probes=50, cg_tol=1e-5, max_cg_iter=1000, seed=999))
    print("snelson exact", exact_mll, "lattice", lattice_mll, "pre_size", pre_size)
    assert abs(lattice_mll - exact_mll) < 0.1


def test_cg_config3_small(plx):
    """Config 3 at reduced N: 50 CG iterations on (s K + sigma^2 I) with [y | 10 probes]; residual falls, one lattice build."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(1234)
    n, d = 100_000, 8
    x = torch.randn(n, d, generator=g).cuda()
    y = torch.randn(n, generator=g).cuda()
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    cache = plx.lattice_cache()
    cache.clear()
    misses0 = cache.misses
    with torch.no_grad():
        mm = model.khat_matmul(x)
        Z = (torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1).cuda()
        rhs = torch.cat([y[:, None], Z], 1)
        sol, info = solvers.batched_cg(mm, rhs, max_iter=50, tol=1e-8)
        resid = (mm(sol) - rhs).norm(dim=0) / rhs.norm(dim=0)
    assert info["iterations"] == 50 and cache.misses == misses0 + 1
    assert float(resid.max()) < 0.05


def test_lattice_row_order_mode(plx):
    """plx_set_row_order: rows in lattice order in and out == permuted caller-order result."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(20000, 4, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice().build(x, taps)
    for vd in (1, 3, 7, 40, 130):       # scan, lane-group and wide splat; single-column, narrow, general and wide blur
        v = torch.randn(20000, vd, generator=g).cuda()
        want = lat.apply(v).clone()
        lat.set_lattice_row_order(True)
        got_l = lat.apply(lat.to_lattice_order(v).contiguous())
        lat.set_lattice_row_order(False)
        assert torch.equal(lat.from_lattice_order(got_l), want)
    # the solver's lattice-order CG gives the same solution as the caller-order CG
    from simplex_gp_amd import solvers
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=4)).cuda()
    rhs = torch.randn(20000, 3, generator=g).cuda()
    with torch.no_grad():
        a, _ = model.khat_solve(x, rhs, max_iter=30, tol=1e-10)
        b, _ = solvers.batched_cg(model.khat_matmul(x), rhs, max_iter=30, tol=1e-10)
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4


def test_coldot_matches_torch(plx):
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(4)
    for n, t in [(1, 1), (1000, 3), (100003, 11), (50000, 64), (20000, 200)]:
        a = torch.randn(n, t, generator=g).cuda()
        b = torch.randn(n, t, generator=g).cuda()
        got = solvers._colsum(a, b)
        want = (a.double() * b.double()).sum(0)
        assert got.shape == (t,)
        assert torch.allclose(got.double(), want, rtol=1e-4, atol=1e-3 * (n ** 0.5)), (n, t)
        assert torch.equal(got, solvers._colsum(a, b))          # deterministic


@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("vertex_order", [0, 2])
def test_sharded_build_equals_replicated_build(plx, shards, vertex_order):
    """plx_build_local + key exchange + plx_build_merge (each rank sees only its rows) against plx_build(shard_index,
    n_shards) on all rows: the same vertex SET and the same per-point structure (corner keys, weights); along the Morton
    curve -- a function of the vertex set -- also the same vertex numbering and neighbour table.  (The point order inside
    a shard is each build's own: its sort keys are laid out over the coordinate ranges that build has seen.)"""
    from simplex_gp_amd import _native as nv
    nv.check(nv.lib().plx_tune(b"vertex_order", vertex_order), "plx_tune")
    try:
        _sharded_equals_replicated(plx, shards, "morton" if vertex_order else "first_touch")
    finally:
        nv.check(nv.lib().plx_tune(b"vertex_order", 1), "plx_tune")


def _sharded_equals_replicated(plx, shards, numbering):
    from simplex_gp_amd import _native as nv
    from simplex_gp_amd.distributed import shard_bounds
    g = torch.Generator().manual_seed(21)
    n, d, vd = 30011, 5, 3
    x = torch.randn(n, d, generator=g).cuda()
    v = torch.randn(n, vd, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    bounds = [shard_bounds(n, shards, r) for r in range(shards)]
    locals_ = [plx.Lattice() for _ in range(shards)]
    keys = [lat.build_local(x[lo:hi].contiguous(), taps) for lat, (lo, hi) in zip(locals_, bounds)]
    counts = [k.shape[0] for k in keys]
    all_keys = torch.cat(keys, 0)
    total = None
    for r, (lat, (lo, hi)) in enumerate(zip(locals_, bounds)):
        lat.build_merge(all_keys, counts, r, total_points=n)
        rep = plx.Lattice().build(x, taps, shard=(r, shards))
        assert lat.m == rep.m and lat.n == hi - lo and lat.n_owned == hi - lo
        assert lat.stage_kernels()["vertex_order"] == [numbering] == rep.stage_kernels()["vertex_order"]
        keys_l, keys_r = lat.export(nv.ARRAY_KEYS), rep.export(nv.ARRAY_KEYS)
        if numbering == "morton":
            assert np.array_equal(keys_l, keys_r)                                              # same numbering
            assert np.array_equal(lat.export(nv.ARRAY_NEIGHBORS), rep.export(nv.ARRAY_NEIGHBORS))
        else:                                                                                  # same set
            assert np.array_equal(np.unique(keys_l, axis=0), np.unique(keys_r, axis=0))
        # per point, in the caller's row order: the d+1 corner keys and weights
        perm_l = lat.export(nv.ARRAY_POINT_PERM).astype(np.int64)
        perm_r = rep.export(nv.ARRAY_POINT_PERM).astype(np.int64)
        ev_l, ev_r = lat.export(nv.ARRAY_ENTRY_VERTEX), rep.export(nv.ARRAY_ENTRY_VERTEX)
        ew_l, ew_r = lat.export(nv.ARRAY_ENTRY_WEIGHT), rep.export(nv.ARRAY_ENTRY_WEIGHT)
        own_r = (perm_r >= lo) & (perm_r < hi)                  # the replicated build holds every row; this shard's are [lo, hi)
        inv_l = np.empty(hi - lo, np.int64); inv_l[perm_l] = np.arange(hi - lo)
        pos_r = np.nonzero(own_r)[0]
        inv_r = np.empty(hi - lo, np.int64); inv_r[perm_r[pos_r] - lo] = pos_r
        assert np.array_equal(keys_l[ev_l[:, inv_l]], keys_r[ev_r[:, inv_r]])
        assert np.array_equal(ew_l[:, inv_l], ew_r[:, inv_r])
        part, part_rep = lat.splat(v[lo:hi]), rep.splat(v[lo:hi])
        if numbering == "morton":
            assert rel_l2(part.cpu().numpy(), part_rep.cpu().numpy()) <= 1e-6
        total = part.clone() if total is None else total + part
        rep.close()
    full = plx.Lattice().build(x, taps).apply(v)
    got = torch.empty_like(full)
    for lat, (lo, hi) in zip(locals_, bounds):
        got[lo:hi] = lat.slice(lat.blur(total.clone(), vd=vd), vd=vd)
    assert rel_l2(got.cpu().numpy(), full.cpu().numpy()) <= 1e-6
    # protocol errors are reported, not UB
    from simplex_gp_amd._native import PlxError
    fresh = plx.Lattice()
    with pytest.raises(PlxError):
        fresh.build_merge(all_keys, counts, 0)            # merge without a local stage
    fresh.build_local(x[:100].contiguous(), taps)
    with pytest.raises(PlxError):
        fresh.build_merge(all_keys, counts, 0)            # announced count does not match


def test_fit_and_predict_on_gpu(plx, tmp_path):
    """The training loop of simplex_gp_amd.training on the HIP path: validation RMSE well below the trivial predictor."""
    from simplex_gp_amd import solvers, training
    g = torch.Generator().manual_seed(2)
    n = 20000
    x = torch.randn(n, 3, generator=g)
    y = torch.sin(2 * x[:, 0]) * torch.cos(x[:, 1]) + 0.3 * x[:, 2] + 0.1 * torch.randn(n, generator=g)
    x, y = x.cuda(), y.cuda()
    tr, va, te = slice(0, 12800), slice(12800, 16000), slice(16000, n)
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=3), min_noise=1e-3).cuda()
    history, best = training.fit(model, (x[tr], y[tr]), val=(x[va], y[va]), test=(x[te], y[te]), epochs=15, lr=0.1,
                                 cg_tol=1.0, checkpoint=str(tmp_path / "model.pt"))
    assert best["summary"]["val/rmse"] < 0.5 * float(y[va].std())
    assert np.isfinite(best["summary"]["test/nll"])
    mean, var = training.predict(model, x[tr], y[tr], x[te])
    assert mean.shape == (n - 16000,) and (var > 0).all()


def test_side_stream_and_buffer_reuse_soak(plx):
    """Work is enqueued on torch's CURRENT stream (not the legacy default one), and one lattice object survives
    a long sequence of rebuilds with growing / shrinking shapes, dimensions, orders and column counts."""
    rng = np.random.default_rng(11)
    lat = plx.Lattice()
    side = torch.cuda.Stream()
    shapes = [(5000, 3, 1, 1), (200, 7, 4, 2), (60000, 2, 1, 1), (1, 5, 3, 0), (3000, 12, 2, 3), (40000, 4, 11, 1),
              (17, 1, 1, 1), (20000, 6, 1, 2), (999, 9, 5, 1), (50000, 3, 1, 1)]
    for n, d, vd, order in shapes:
        ref = (rng.standard_normal((n, d)) * rng.choice([0.3, 1.0, 3.0])).astype(np.float32)
        src = rng.standard_normal((n, vd)).astype(np.float32)
        taps = np.array([0.1, 0.3, 0.6, 1.0, 0.6, 0.3, 0.1][3 - order: 4 + order], np.float32)
        ref_t, src_t = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            lat.build(ref_t, taps)
            out = lat.apply(src_t)
            out2 = lat.apply(src_t)          # second MVM on the same lattice
        side.synchronize()
        oracle.set_exact_mode(False)
        want, m = oracle.filter(src, ref, taps, return_m=True)
        oracle.set_exact_mode(True)
        assert lat.m == m, (n, d, vd, order)
        assert rel_l2(out.cpu().numpy(), want) <= 5e-5, (n, d, vd, order)
        assert torch.equal(out, out2)
    lat.close()


def test_buffers_survive_the_stream_they_grew_on(plx):
    """Device buffers grow stream-ordered on the stream of the call that needs them.  A lattice built on a stream its
    owner then DESTROYS must still rebuild (bigger: every buffer is re-allocated) and serve MVMs on another stream, and
    a lattice may change streams between calls as long as the caller orders them."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(raw)) == 0
    rng = np.random.default_rng(23)
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice()

    def case(n, d, vd, stream):
        ref = rng.standard_normal((n, d)).astype(np.float32)
        src = rng.standard_normal((n, vd)).astype(np.float32)
        ref_t, src_t = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            lat.build(ref_t, taps)
            out = lat.apply(src_t)
        stream.synchronize()
        oracle.set_exact_mode(False)
        want = oracle.filter(src, ref, taps)
        oracle.set_exact_mode(True)
        assert rel_l2(out.cpu().numpy(), want) <= 5e-5, (n, d, vd)
        return ref_t, src_t, out

    ext = torch.cuda.ExternalStream(raw.value)
    case(3000, 3, 1, ext)
    case(2000, 5, 4, ext)
    del ext
    assert hip.hipStreamDestroy(raw) == 0                      # the owner of every buffer of `lat` is gone
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    ref_t, src_t, out_a = case(40000, 6, 1, a)                 # everything grows: the old owner cannot be waited for
    # same lattice, another stream, ordered by the caller; a wider right-hand side grows the workspace on stream b
    wide = torch.from_numpy(rng.standard_normal((40000, 7)).astype(np.float32)).cuda()
    torch.cuda.synchronize()
    b.wait_stream(a)
    with torch.cuda.stream(b):
        out_b = lat.apply(src_t)
        out_w = lat.apply(wide)
    b.synchronize()
    assert torch.equal(out_a, out_b)
    with torch.cuda.stream(a):
        a.wait_stream(b)
        first = lat.apply(wide[:, :1].contiguous())
    a.synchronize()
    assert rel_l2(first.cpu().numpy(), out_w[:, :1].cpu().numpy()) <= 1e-5      # (one-column and row kernels: two summation orders)
    lat.close()


def test_mvm_is_graph_capturable(plx):
    """Once a lattice's tables exist and its workspace has been sized, an MVM neither synchronises nor allocates inside
    the library: it can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed on new right-hand sides."""
    rng = np.random.default_rng(31)
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    n, d = 50000, 5
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    lat = plx.Lattice().build(x, taps)
    for vd in (1, 4):
        v = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).cuda()
        v2 = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).cuda()
        out = torch.empty_like(v)
        lat.prepare(vd)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                lat.apply(v, out)                      # tables, workspace high-water mark
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            lat.apply(v, out)
        want1, want2 = lat.apply(v).clone(), lat.apply(v2).clone()
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want1), vd
        v.copy_(v2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want2), vd
    lat.close()
    # a capture that would have to BUILD tables (allocation + a host read-back the capture would only record) is refused
    # with the remedy in the message, nothing is launched, and the lattice works eagerly afterwards
    from simplex_gp_amd._native import PlxError
    lat2 = plx.Lattice().build(x, taps)
    v = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).cuda()
    out = torch.empty_like(v)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(PlxError, match="plx_prepare"):
        with torch.cuda.graph(graph):
            lat2.apply(v, out)
    torch.cuda.synchronize()
    want = lat2.apply(v).clone()
    lat3 = plx.Lattice().build(x, taps)
    assert torch.equal(lat3.apply(v), want)
    lat2.close()
    lat3.close()


def test_rejected_build_leaves_a_built_lattice_alone(plx):
    """A build call that fails validation must not touch the lattice: it stays built, with the switch snapshot it was built
    under (plx.h: 'a plx_tune call changes nothing for lattices that are already built')."""
    from simplex_gp_amd import _native as nv
    from simplex_gp_amd._native import PlxError
    rng = np.random.default_rng(5)
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    x = torch.from_numpy(rng.standard_normal((20000, 4)).astype(np.float32)).cuda()
    v = torch.from_numpy(rng.standard_normal((20000, 1)).astype(np.float32)).cuda()
    lat = plx.Lattice().build(x, taps)
    want = lat.apply(v).clone()
    kern = lat.stage_kernels()
    try:
        nv.check(nv.lib().plx_tune(b"block_path", 0), "plx_tune")          # new process default: never the block tables
        with pytest.raises((PlxError, ValueError)):
            lat.build(x, np.array([0.5, 0.5], np.float32))                  # even tap count: rejected
        with pytest.raises(PlxError):
            lat.build(x[:0], taps)                                          # n = 0: rejected inside the library
        assert lat.m > 0 and torch.equal(lat.apply(v), want) and lat.stage_kernels() == kern
    finally:
        nv.check(nv.lib().plx_tune(b"block_path", 1), "plx_tune")
    lat.close()


def test_two_host_threads_two_lattices(plx):
    """The library keeps no mutable process state besides the plx_tune defaults: two host threads, each with its own
    lattice and stream, rebuild and apply concurrently (ctypes releases the GIL inside the calls) and reproduce, bit
    for bit, what each computes alone."""
    import threading
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    jobs = []
    for seed, (n, d, vd) in enumerate([(60000, 4, 1), (30000, 8, 3)]):
        rng = np.random.default_rng(100 + seed)
        refs = [torch.from_numpy((rng.standard_normal((n + 1000 * k, d)) / (1.0 + 0.5 * k)).astype(np.float32)).cuda() for k in range(3)]
        srcs = [torch.from_numpy(rng.standard_normal((r.shape[0], vd)).astype(np.float32)).cuda() for r in refs]
        jobs.append((refs, srcs))
    torch.cuda.synchronize()

    def run(job, rounds, out):
        refs, srcs = job
        lat, stream = plx.Lattice(), torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for it in range(rounds):
                k = it % len(refs)
                lat.build(refs[k], taps)
                res = lat.apply(srcs[k])
                res = lat.apply(srcs[k])
                out.append(res.clone())
        stream.synchronize()
        lat.close()

    alone = [[], []]
    for j in range(2):
        run(jobs[j], 3, alone[j])
    together = [[], []]
    threads = [threading.Thread(target=run, args=(jobs[j], 12, together[j])) for j in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for j in range(2):
        assert len(together[j]) == 12
        for it, res in enumerate(together[j]):
            assert torch.equal(res, alone[j][it % 3]), (j, it)


def test_fused_cg_updates_match_torch(plx):
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(6)
    for n, t in [(1000, 1), (100003, 11), (5000, 40)]:
        X, R, P, AP = (torch.randn(n, t, generator=g).cuda() for _ in range(4))
        alpha = torch.randn(t, generator=g).cuda()
        beta = torch.randn(t, generator=g).cuda()
        Xw, Rw = X + P * alpha, R - AP * alpha
        rs = solvers._cg_update(X, R, P, AP, alpha)
        assert torch.allclose(X, Xw, atol=1e-6) and torch.allclose(R, Rw, atol=1e-6)
        assert torch.allclose(rs.double(), (Rw.double() ** 2).sum(0), rtol=1e-4)
        Pw = R + P * beta
        solvers._cg_direction(P, R, beta)
        assert torch.allclose(P, Pw, atol=1e-6)


def test_preconditioned_solve_on_gpu(plx):
    """Pivoted-Cholesky preconditioner on the HIP path (rows of K through one-hot MVMs): same solution as plain CG
    in fewer iterations at small noise, and the preconditioned MLL agrees with the unpreconditioned one."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(5)
    n, d = 30000, 2
    x = torch.randn(n, d, generator=g).cuda()
    y = (torch.sin(2 * x[:, 0]) + 0.05 * torch.randn(n, generator=g).cuda())
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
    with torch.no_grad():
        model.raw_noise.fill_(-5.0)
        rhs = y[:, None].contiguous()
        sol0, info0 = model.khat_solve(x, rhs, max_iter=2000, tol=1e-4, check_every=1)
        pre = model.preconditioner(x, 50)
        sol1, info1 = model.khat_solve(x, rhs, max_iter=2000, tol=1e-4, check_every=1, precond=pre)
        mm = model.khat_matmul(x)
        r0 = float((mm(sol0) - rhs).norm() / rhs.norm())
        r1 = float((mm(sol1) - rhs).norm() / rhs.norm())
    print("cg iterations", info0["iterations"], "pcg", info1["iterations"], "residuals", r0, r1)
    assert r0 < 2e-3 and r1 < 2e-3
    assert info1["iterations"] < info0["iterations"]
    a = solvers.marginal_log_likelihood(model, x, y, num_probes=20, cg_tol=1e-3, seed=0)
    b = solvers.marginal_log_likelihood(model, x, y, num_probes=20, cg_tol=1e-3, seed=0, pre_size=50)
    b.backward()
    assert abs(float(a.detach()) - float(b.detach())) < 0.02 * (1 + abs(float(a.detach())))
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("n,d,L", [(3000, 8, 11), (2000, 18, 11), (1500, 8, 7), (2501, 3, 28)])
def test_fused_backward_matches_unfused_and_oracle(plx, n, d, L):
    """plx_apply_backward (stack formed inside the splat, slice + contraction in one kernel) against (a) the
    three-call native form and (b) the reference formulation (py:113-123) over the CPU oracle filter."""
    g = torch.Generator().manual_seed(100 + d + L)
    x0 = (torch.randn(n, d, generator=g) * 0.7)
    v0 = torch.randn(n, L, generator=g)
    w0 = torch.randn(n, L, generator=g)
    dk = plx.DiscretizedKernelFN(plx.rbf, 1)
    assert plx.Lattice.backward_fusable(L, d)

    def grads(device):
        x = x0.to(device).requires_grad_(True)
        v = v0.to(device).requires_grad_(True)
        out = plx.LatticeFilterGeneral.apply(v, x, dk)
        (out * w0.to(device)).sum().backward()
        return v.grad.cpu().numpy(), x.grad.cpu().numpy()

    assert plx.LatticeFilterGeneral.method is None and plx.LatticeFilterGeneral.fused_backward
    fused = grads("cuda")
    plx.LatticeFilterGeneral.fused_backward = False
    try:
        unfused = grads("cuda")
    finally:
        plx.LatticeFilterGeneral.fused_backward = True

    def oracle_filter(src, ref, coeffs):
        oracle.set_exact_mode(False)
        try:
            return torch.from_numpy(oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy()))
        finally:
            oracle.set_exact_mode(True)
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    try:
        ref = grads("cpu")
    finally:
        plx.LatticeFilterGeneral.method = None
    for name, a, b, c in (("grad_src", fused[0], unfused[0], ref[0]), ("grad_x", fused[1], unfused[1], ref[1])):
        assert rel_l2(a, b) <= 1e-6, (name, "fused vs unfused", rel_l2(a, b))
        assert rel_l2(a, c) <= 2e-5, (name, "fused vs oracle", rel_l2(a, c))


def test_affine_apply_and_device_side_cg(plx):
    """plx_apply_affine == s * apply + noise * v, in both row orders; the CG loop with device-side coefficients
    (plx_cg_step_*) follows the tensor formulation (same iterates to rounding, same Lanczos tridiagonals)."""
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(21)
    n, d, t = 50000, 4, 6
    x = torch.randn(n, d, generator=g).cuda()
    V = torch.randn(n, t, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice().build(x, taps)
    ss = torch.tensor([0.7, 0.3], device="cuda")
    for lattice_rows in (False, True):
        lat.set_lattice_row_order(lattice_rows)
        ref = lat.apply(V) * 0.7 + 0.3 * V
        got = lat.apply_affine(V, ss)
        assert rel_l2(got.cpu().numpy(), ref.cpu().numpy()) <= 1e-6
        one = lat.apply_affine(V[:, :1].contiguous(), ss)
        assert rel_l2(one.cpu().numpy(), ref[:, :1].cpu().numpy()) <= 1e-6
    from simplex_gp_amd._native import PlxError
    with pytest.raises(PlxError):
        lat.apply_affine(V, ss, out=V)
    lat.set_lattice_row_order(False)
    # the fused column dot <v, out> (plx_apply_affine_dot) against a separate reduction, widths with and without padding
    for cols in (2, 3, 4, 6):
        Vc = V[:, :cols].contiguous()
        o2, dot = lat.apply_affine(Vc, ss, want_dot=True)
        assert torch.equal(o2, lat.apply_affine(Vc, ss))
        want = (Vc.double() * o2.double()).sum(0)
        assert torch.allclose(dot.double(), want, rtol=1e-4, atol=1e-3), cols
    with pytest.raises(ValueError):
        lat.apply_affine(V[:, :1].contiguous(), ss, want_dot=True)
    mm = lambda W: lat.apply_affine(W, ss)       # noqa: E731
    Xa, ia = solvers.batched_cg(mm, V, max_iter=40, tol=1e-6, want_tridiag=True, check_every=1)
    Xc, ic = solvers.batched_cg(mm, V, max_iter=40, tol=1e-6, want_tridiag=True, check_every=1,
                                matmul_dot=lambda W: lat.apply_affine(W, ss, want_dot=True))
    # a different summation order of p^T A p: the fp32 CG trajectories drift apart by a few 1e-5 over 40 iterations
    assert ic["iterations"] == ia["iterations"] and rel_l2(Xc.cpu().numpy(), Xa.cpu().numpy()) <= 5e-4
    ra, rc = (mm(Xa) - V).norm() / V.norm(), (mm(Xc) - V).norm() / V.norm()
    assert float(rc) <= 1.5 * float(ra) + 1e-6
    Xb, ib = solvers.batched_cg(mm, V, max_iter=40, tol=1e-6, want_tridiag=True, check_every=1, reduce=lambda s: s)
    assert ia["iterations"] == ib["iterations"]
    assert rel_l2(Xa.cpu().numpy(), Xb.cpu().numpy()) <= 1e-5
    k = min(ia["tridiag"].shape[1], ib["tridiag"].shape[1])
    assert torch.allclose(ia["tridiag"][:, :k, :k], ib["tridiag"][:, :k, :k], rtol=1e-3, atol=1e-4)
    lat.close()


@pytest.mark.parametrize("t", [4, 8, 12, 16])
def test_cg_iteration_without_standalone_reductions(plx, t):
    """plx_cg_step_update_fused / plx_cg_step_direction_fused (round 6: the <P, AP> partial sums of the slice kernel and the
    |R|^2 partial sums of the update are added up inside their consumers, two launches fewer per iteration) against the
    pair with stand-alone reductions: one step on random vectors, coefficient by coefficient; then whole solves -- the
    same iteration counts, iterates equal to rounding, the same Lanczos tridiagonals; reproducible bit for bit."""
    import ctypes
    from simplex_gp_amd import solvers, _native as nv
    lib = nv.lib()
    g = torch.Generator().manual_seed(40 + t)
    n, d = 60001, 4
    x = torch.randn(n, d, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice().build(x, taps)
    lat.set_lattice_row_order(True)
    ss = torch.tensor([0.7, 0.3], device="cuda")
    p = lambda a: ctypes.c_void_p(a.data_ptr())          # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # ---- one step
    X, R, P = (torch.randn(n, t, generator=g).cuda() for _ in range(3))
    rs = (R * R).sum(0).contiguous()
    active = torch.ones(t, device="cuda")
    active[1] = 0.0                                                       # a frozen column: alpha = beta = 0 there
    b_norm = rs.sqrt().contiguous()
    AP, dot = lat.apply_affine(P, ss, want_dot=True)
    X0, R0, P0 = X.clone(), R.clone(), P.clone()
    work = torch.empty(int(lib.plx_coldot_work_floats(t)), device="cuda")
    rs_new, alpha, beta, act2 = (torch.empty(t, device="cuda") for _ in range(4))
    nv.check(lib.plx_cg_step_update(p(X0), p(R0), p(P0), p(AP), p(rs), p(dot.contiguous()), p(active), n, t, p(rs_new), p(alpha), p(work), stream), "update")
    nv.check(lib.plx_cg_step_direction(p(P0), p(R0), p(rs_new), p(rs), p(active), p(b_norm), 1e-3, n, t, p(beta), p(act2), stream), "direction")
    AP2, part, tiles = lat.apply_affine(P, ss, want_dot="partial")
    assert torch.equal(AP2, AP) and tiles == int(lib.plx_affine_dot_tiles(lat._h, t)) > 0
    X1, R1, P1 = X.clone(), R.clone(), P.clone()
    fwork = torch.empty(int(lib.plx_cg_fused_work_floats(t)), device="cuda")
    rs_new1, alpha1, beta1, act21 = (torch.empty(t, device="cuda") for _ in range(4))
    nv.check(lib.plx_cg_step_update_fused(p(X1), p(R1), p(P1), p(AP2), p(rs), p(part), tiles, p(active), n, t, p(alpha1), p(fwork), stream), "update_fused")
    nv.check(lib.plx_cg_step_direction_fused(p(P1), p(R1), p(fwork), p(rs), p(active), p(b_norm), 1e-3, n, t, p(rs_new1), p(beta1), p(act21), stream), "direction_fused")
    assert float(alpha1[1]) == 0.0 and float(beta1[1]) == 0.0 and torch.equal(act2, act21)
    assert torch.allclose(alpha1, alpha, rtol=2e-5) and torch.allclose(rs_new1, rs_new, rtol=2e-5) and torch.allclose(beta1, beta, rtol=4e-5)
    assert torch.allclose(rs_new1.double(), (R1.double() ** 2).sum(0), rtol=1e-5)
    for a, b in ((X1, X0), (R1, R0), (P1, P0)):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5
    assert int(lib.plx_cg_fused_work_floats(11)) == -1
    # ---- whole solves through solvers.batched_cg (the path khat_solve takes)
    V = torch.randn(n, t, generator=g).cuda()

    def mm_dot(W):
        return lat.apply_affine(W, ss, want_dot=True)
    mm_dot.partial = lambda W: lat.apply_affine(W, ss, want_dot="partial")
    mm = lambda W: lat.apply_affine(W, ss)       # noqa: E731
    default = solvers.FUSED_CG_STEPS
    assert default == "auto" and solvers._fuse_cg_steps(n) and not solvers._fuse_cg_steps(1_000_000)
    try:
        solvers.FUSED_CG_STEPS = True
        Xf, inf_f = solvers.batched_cg(mm, V, max_iter=20, tol=1e-6, want_tridiag=True, check_every=1, matmul_dot=mm_dot)
        Xf2, _ = solvers.batched_cg(mm, V, max_iter=20, tol=1e-6, want_tridiag=True, check_every=1, matmul_dot=mm_dot)
        assert torch.equal(Xf, Xf2)                                        # fixed summation orders: reproducible
        solvers.FUSED_CG_STEPS = False
        Xu, inf_u = solvers.batched_cg(mm, V, max_iter=20, tol=1e-6, want_tridiag=True, check_every=1, matmul_dot=mm_dot)
    finally:
        solvers.FUSED_CG_STEPS = default
    # (two fp32 CG runs whose dot products are associated differently drift apart like cond(A) x eps x iterations: 6.5e-4
    # measured here at 20 iterations of an unconverged solve; what must hold is that neither is the worse solution -- the
    # true residuals below -- and that the quadrature agrees)
    assert inf_f["iterations"] == inf_u["iterations"] and rel_l2(Xf.cpu().numpy(), Xu.cpu().numpy()) <= 3e-3
    # the Lanczos coefficients: the leading block to rounding (another association of the dot products: the fp32 recurrences
    # drift apart by a few 1e-5 per iteration, and the late coefficients of a converged column are noise in both), and
    # what they are FOR -- the quadrature e1^T log(T) e1 of the log-determinant -- to 1e-3
    Tf, Tu = inf_f["tridiag"].cpu(), inf_u["tridiag"].cpu()
    assert torch.allclose(Tf[:, :6, :6], Tu[:, :6, :6], rtol=5e-3, atol=5e-4)
    qf, qu = solvers.slq_terms(Tf), solvers.slq_terms(Tu)
    assert torch.allclose(qf, qu, rtol=2e-3, atol=2e-4), (qf, qu)
    rf, ru = (mm(Xf) - V).norm() / V.norm(), (mm(Xu) - V).norm() / V.norm()
    assert float(rf) <= 1.5 * float(ru) + 1e-6
    lat.close()


@pytest.mark.parametrize("n,d,ell,order,vd", [(3000, 18, 1.0, 3, 418), (3000, 18, 1.0, 1, 130), (20000, 8, 0.2, 1, 198),
                                              (20000, 8, 0.2, 2, 256), (5000, 12, 0.5, 3, 500), (40000, 4, 0.05, 1, 126),
                                              (3000, 18, 1.0, 3, 101), (20000, 8, 0.2, 1, 68), (5000, 12, 0.5, 2, 124)])
def test_wide_blur_on_active_rows_equals_the_dense_passes(plx, n, d, ell, order, vd):
    """Wide rows on sparse lattices (round 6): a blur pass that touches only the vertices with a neighbour on its axis, in place
    (blur_active_rows_kernel + blur_active_store_kernel; the centre tap of every kernel profile is exactly 1, so the other rows
    do not change), gives what the dense passes give -- the same operations in the same order for the rows that change, the
    untouched rows as they were -- on lattices where every point has a simplex of its own and on partly shared ones, orders
    1-3, one and two chunks per lane, rows of 17-31 chunks (lanes idle: the evaluation's 101 columns).  Also: the lists are rebuilt with the lattice, a centre tap other than 1 keeps the
    dense passes, and the whole filter (splat, blur, slice) agrees with the oracle."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    g = torch.Generator().manual_seed(n + d + vd)
    x = (torch.randn(n, d, generator=g) / ell).contiguous()
    taps = {1: [0.34608543, 1.0, 0.34608543], 2: [0.0826, 0.5362, 1.0, 0.5362, 0.0826],
            3: [0.0844, 0.2424, 0.6031, 1.0, 0.6031, 0.2424, 0.0844]}[order]
    taps = np.array(taps, np.float32)
    lat = plx.Lattice().build(x.cuda(), taps)
    m = lat.m
    vals = torch.randn(m, lat.values_stride(vd), generator=g).cuda()
    outs, kinds = {}, {}
    try:
        for mode in (0, 2, 1):
            nv.check(lib.plx_tune(b"blur_active", mode), "plx_tune")
            lat.build(x.cuda(), taps)
            a, b = vals.clone(), torch.empty_like(vals)
            outs[mode] = lat.blur(a, b, vd=vd).clone()
            kinds[mode] = lat.stage_kernels()["blur_axis"]
    finally:
        nv.check(lib.plx_tune(b"blur_active", 1), "plx_tune")
    assert kinds[0] == ["blur_axis_multi_kernel"] and kinds[2] == ["blur_active_rows_kernel", "blur_active_store_kernel"], kinds
    assert torch.equal(outs[2], outs[0])                       # (numeric equality: a zero may change its sign, nothing else may differ)
    assert torch.equal(outs[1], outs[0])
    nbr = lat.export(nv.ARRAY_NEIGHBORS)
    share = float((nbr >= 0).any(axis=1).mean())
    sparse = m >= 0.75 * n * (d + 1) and share <= 0.40
    assert (kinds[1] == kinds[2]) == sparse, (kinds[1], share, m / (n * (d + 1)))
    # a centre tap that is not 1: the rows without neighbours DO change (scaled): the dense passes run whatever the switch says
    odd = taps.copy()
    odd[order] = 0.9
    try:
        nv.check(lib.plx_tune(b"blur_active", 2), "plx_tune")
        lat.build(x.cuda(), odd)                      # (other taps, other scale factors: another lattice, another m)
        vals2 = torch.randn(lat.m, lat.values_stride(vd), generator=g).cuda()
        forced = lat.blur(vals2.clone(), torch.empty_like(vals2), vd=vd).clone()
        assert lat.stage_kernels()["blur_axis"] == ["blur_axis_multi_kernel"]
        nv.check(lib.plx_tune(b"blur_active", 0), "plx_tune")
        lat.build(x.cuda(), odd)
        assert lat.m == vals2.shape[0]
        assert torch.equal(forced, lat.blur(vals2.clone(), torch.empty_like(vals2), vd=vd))
    finally:
        nv.check(lib.plx_tune(b"blur_active", 1), "plx_tune")
    # the whole filter against the oracle on this lattice (whatever path the gate picks)
    lat.build(x.cuda(), taps)
    src = torch.randn(n, vd, generator=g)
    oracle.set_exact_mode(False)
    try:
        want = oracle.filter(src.numpy(), x.numpy(), taps)
    finally:
        oracle.set_exact_mode(True)
    got = lat.apply(src.cuda()).cpu().numpy()
    assert rel_l2(got, want) <= 1e-5
    lat.close()


def test_point_order_warm_start_across_rescaled_rebuilds(plx):
    """Lattice.build(reuse_order=True) / plx_set_reuse_order: a rebuild on re-scaled positions keeps the point order of the
    previous build (the order passes are skipped) and gives the cold build's structure and output (up to the order of the
    fp32 sums inside a vertex row); it is ignored for another row count.  The kernel-level cache does it by itself when the
    SAME data tensor comes back under another lengthscale (LatticeAccelerated.forward tags its scaled positions), not for
    other data, and refreshes the order after MAX_ORDER_AGE rebuilds."""
    from simplex_gp_amd import lattice_kernel as lk, _native as nv
    g = torch.Generator().manual_seed(77)
    n, d = 40000, 5
    x = torch.randn(n, d, generator=g).cuda()
    v = torch.randn(n, 3, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    cold, warm = plx.Lattice(), plx.Lattice()
    assert warm.order_age == -1
    warm.build((x / 0.7).contiguous(), taps)
    assert warm.order_age == 0
    perm0 = warm.export(nv.ARRAY_POINT_PERM).copy()
    ref1 = (x / 0.9).contiguous()
    warm.build(ref1, taps, reuse_order=True)
    cold.build(ref1, taps)
    assert warm.order_age == 1 and cold.order_age == 0
    assert np.array_equal(warm.export(nv.ARRAY_POINT_PERM), perm0)                 # the order was kept ...
    assert not np.array_equal(cold.export(nv.ARRAY_POINT_PERM), perm0)             # ... where a cold build finds another
    assert warm.m == cold.m
    kw, kc = warm.export(nv.ARRAY_KEYS), cold.export(nv.ARRAY_KEYS)
    assert {k.tobytes() for k in kw} == {k.tobytes() for k in kc}                  # the same vertex set
    assert rel_l2(warm.apply(v).cpu().numpy(), cold.apply(v).cpu().numpy()) <= 1e-6
    warm.build(ref1[: n // 2].contiguous(), taps, reuse_order=True)                # other rows: the flag is ignored
    assert warm.order_age == 0
    warm.build(ref1, taps)                                                         # one shot: a plain build is a cold one
    assert warm.order_age == 0
    cold.close(); warm.close()
    # ---- the cache: same data tensor, lengthscale moved
    cache = lk.lattice_cache()
    cache.clear()
    k = plx.RBFLattice(order=1, ard_num_dims=d).cuda()
    w0 = cache.warm_rebuilds
    outs = []
    with torch.no_grad():
        for step, ell in enumerate([0.7, 0.75, 0.8]):
            k.lengthscale = ell
            outs.append(k(x, x).matmul(v))
            assert cache.warm_rebuilds == w0 + step, (step, cache.warm_rebuilds)
            assert len(cache._entries) == 1                                        # rebuilt in place, not piled up
        lat = list(cache._entries.values())[0][0]
        assert lat.order_age == 2
        want = plx.Lattice().build((x / 0.8).contiguous(), taps).apply(v)
        assert rel_l2(outs[-1].cpu().numpy(), want.cpu().numpy()) <= 1e-6
        x2 = x.clone()                                                             # other data (another tensor): no warm start
        k(x2, x2).matmul(v)
        assert cache.warm_rebuilds == w0 + 2 and len(cache._entries) == 2
        seen = []
        for i in range(lk.MAX_ORDER_AGE + 2):                                      # the order is refreshed now and then
            k.lengthscale = 0.8 + 0.01 * (i + 1)
            k(x, x).matmul(v)
            seen.append(max(e[0].order_age for e in cache._entries.values()))
        assert max(seen) == lk.MAX_ORDER_AGE and seen[-1] < lk.MAX_ORDER_AGE and len(cache._entries) == 2
    cache.clear()


def test_fused_entry_points_reject_bad_use(plx):
    """plx_apply_backward / plx_apply_affine fail loudly (error code -> PlxError / ValueError), never silently."""
    from simplex_gp_amd._native import PlxError
    g = torch.Generator().manual_seed(8)
    n, d, L = 4000, 8, 11
    x = torch.randn(n, d, generator=g).cuda()
    v = torch.randn(n, L, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice()
    with pytest.raises((PlxError, ValueError)):         # not built yet
        lat.apply_backward(v, v, x)
    lat.build(x, taps, shard=(0, 2))                    # a sharded lattice needs the all-reduce between splat and blur
    with pytest.raises(PlxError):
        lat.apply_backward(v[: lat.n_owned], v[: lat.n_owned], x[: lat.n_owned])
    lat.build(x, taps)
    with pytest.raises(PlxError):                       # 2 * 1 * 9 = 18 columns: outside the fused range
        lat.apply_backward(v[:, :1].contiguous(), v[:, :1].contiguous(), x)
    with pytest.raises(ValueError):                     # shapes must agree
        lat.apply_backward(v, v[:, :5].contiguous(), x)
    with pytest.raises((TypeError, ValueError)):
        lat.apply_affine(v, torch.tensor([1.0, 0.0]))   # scale/shift must live on the device
    assert not plx.Lattice.backward_fusable(1, 8) and plx.Lattice.backward_fusable(11, 8)
    gr, gs = lat.apply_backward(v, v, x, want_grad_src=False)
    assert gs is None and gr.shape == (n, d) and torch.isfinite(gr).all()
    lat.close()


@pytest.mark.parametrize("cname", sorted(CASES))
def test_torch_extension_as_reference_method(plx, golden_dir, cname):
    """The swap a reference maintainer makes: LatticeFilterGeneral.method = <compiled extension>.filter
    (bilateral_kernel.py:60, py:94-95), then the reference-generated autograd goldens through it."""
    ext = plx.torch_ext.load()
    host = np.load(os.path.join(golden_dir, "host_side.npz"))
    profiles = {"rbf": plx.rbf, "matern15": lambda d2: plx.Matern.apply(d2, 1.5)}
    pname, order = CASES[cname]
    dk = plx.DiscretizedKernelFN(profiles[pname], order)
    x = torch.from_numpy(host[f"autograd/{cname}/x"]).cuda().requires_grad_(True)
    s = torch.from_numpy(host[f"autograd/{cname}/src"]).cuda().requires_grad_(True)
    gout = torch.from_numpy(host[f"autograd/{cname}/grad_out"]).cuda()
    plx.LatticeFilterGeneral.method = staticmethod(ext.filter)
    try:
        out = plx.LatticeFilterGeneral.apply(s, x, dk)
        out.backward(gout)
    finally:
        plx.LatticeFilterGeneral.method = None
    for got, name in [(out, "out"), (s.grad, "grad_src"), (x.grad, "grad_x")]:
        assert rel_l2(got.detach().cpu().numpy(), host[f"autograd/{cname}/{name}"]) <= 2e-5, name


def test_torch_extension_matches_ctypes_path(plx):
    """filter() and the staged LatticeHandle of the extension against the ctypes path: same library, same bits; work is
    enqueued on torch's current stream; non-contiguous sources and bad inputs behave like the reference's checks."""
    ext = plx.torch_ext.load()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(30000, 4, generator=g).cuda()
    v = torch.randn(30000, 3, generator=g).cuda()
    taps = torch.tensor([0.34608543, 1.0, 0.34608543])
    want = plx.filter(v, x, taps)
    assert torch.equal(ext.filter(v, x, taps), want)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        got = ext.filter(v, x, taps)
    side.synchronize()
    assert torch.equal(got, want)
    vt = v.t().contiguous().t()                               # non-contiguous source: accepted (py:95 passes it as is)
    assert not vt.is_contiguous() and torch.equal(ext.filter(vt, x, taps), want)
    h = ext.LatticeHandle(0)
    h.build(x, taps)
    assert h.num_vertices == plx.Lattice().build(x, taps.numpy()).m
    assert torch.equal(h.apply(v), want) and torch.equal(h.apply(v[:, :1].contiguous()), plx.filter(v[:, :1].contiguous(), x, taps))
    with pytest.raises(RuntimeError, match="Incompatible shapes"):
        ext.filter(v[:5], x, taps)
    with pytest.raises(RuntimeError, match="float32"):
        ext.filter(v.double(), x, taps)
    with pytest.raises(RuntimeError, match="odd number of taps"):
        ext.filter(v, x, torch.tensor([0.5, 0.5]))


def test_radix_sort_selftest(plx):
    """The build's own stable radix sort (plx_radix.h; rocPRIM above 3e6 64-bit keys) on keys with long runs of
    duplicates: ascending, equal keys in input order, values still with their keys -- at the sizes and bit counts the
    build uses (point order, vertex Morton codes, block rows, CSR) and at the edges (1 key, tile boundaries, one bit)."""
    import ctypes
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cases = [(1, 8, 36), (2, 4, 19), (2047, 8, 40), (2048, 8, 40), (2049, 4, 1), (4097, 4, 19), (100003, 8, 63), (400000, 8, 40),
             (699999, 8, 37), (700001, 8, 37), (1000000, 8, 36), (2770000, 4, 19), (3000001, 8, 36), (9000000, 4, 21), (1000000, 4, 32),
             (300000, 8, 64)]
    for n, key_bytes, bits in cases:
        bad = ctypes.c_int64(-1)
        nv.check(lib.plx_selftest_sort(n, key_bytes, bits, 1234 + n, stream, ctypes.byref(bad)), "plx_selftest_sort")
        assert bad.value == 0, (n, key_bytes, bits, bad.value)
