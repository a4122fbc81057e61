"""Host-side logic on CPU: taps, autograd formulas, operator surface, C ABI.

The native filter is replaced by the CPU oracle through the reference's own
injection point (LatticeFilterGeneral.method, bilateral_kernel.py:60), so the
Python mirror is checked against goldens captured from the reference's
bilateral_kernel.py without a GPU.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import simplex_gp_amd as plx
from simplex_gp_amd import _native, gp_compat
from oracle import oracle


def oracle_filter(src, ref, coeffs):
    out = oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy())
    return torch.from_numpy(out)


@pytest.fixture
def cpu_method():
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    yield
    plx.LatticeFilterGeneral.method = None


@pytest.fixture(scope="module")
def host(golden_dir):
    return np.load(os.path.join(golden_dir, "host_side.npz"))


PROFILES = {
    "rbf": plx.rbf,
    "matern15": lambda d2: plx.Matern.apply(d2, 1.5),
    "matern25": lambda d2: plx.Matern.apply(d2, 2.5),
}


@pytest.mark.parametrize("pname", sorted(PROFILES))
@pytest.mark.parametrize("order", [1, 2, 3])
def test_taps_match_reference(host, pname, order):
    k = plx.DiscretizedKernelFN(PROFILES[pname], order)
    np.testing.assert_allclose(k.get_coeffs().numpy(), host[f"coeffs/{pname}_o{order}/fwd"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(k.get_deriv_coeffs().numpy(), host[f"coeffs/{pname}_o{order}/deriv"], rtol=0, atol=1e-6)


def test_known_rbf_taps():
    # notebooks/viz_mvm.ipynb:78 prints [0.3461, 1.0000, 0.3461] for RBF order 1
    c = plx.get_coeffs(lambda d: plx.rbf(d ** 2), 1)
    assert abs(float(c[0]) - 0.34608543) < 1e-6 and float(c[1]) == 1.0 and c.shape == (3,)


def test_matern_profiles_agree_and_gradient():
    d2 = torch.linspace(0.01, 9, 50, dtype=torch.float64, requires_grad=True)
    for nu in (1.5, 2.5):
        a = plx.Matern.apply(d2, nu)
        b = plx.matern(d2, nu)
        assert torch.allclose(a, b)
        (ga,) = torch.autograd.grad(a.sum(), d2)
        (gb,) = torch.autograd.grad(b.sum(), d2)
        assert torch.allclose(ga, gb, atol=1e-10)
    with pytest.raises(NotImplementedError):
        plx.matern(d2, 0.7)


CASES = {
    "n50_d3_L2_rbf_o1": ("rbf", 1), "n200_d1_L1_rbf_o1": ("rbf", 1),
    "n50_d3_L2_matern15_o3": ("matern15", 3), "n200_d1_L1_matern15_o3": ("matern15", 3),
    "n300_d4_L3_rbf_o2": ("rbf", 2),
}


@pytest.mark.parametrize("cname", sorted(CASES))
def test_autograd_matches_reference(host, cpu_method, cname):
    pname, order = CASES[cname]
    dk = plx.DiscretizedKernelFN(PROFILES[pname], order)
    x = torch.from_numpy(host[f"autograd/{cname}/x"]).requires_grad_(True)
    s = torch.from_numpy(host[f"autograd/{cname}/src"]).requires_grad_(True)
    gout = torch.from_numpy(host[f"autograd/{cname}/grad_out"])
    out = plx.LatticeFilterGeneral.apply(s, x, dk)
    out.backward(gout)

    def close(a, name, tol=2e-5):
        b = host[f"autograd/{cname}/{name}"]
        err = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert err <= tol, (name, err)

    close(out.detach().numpy(), "out")
    close(s.grad.numpy(), "grad_src")
    close(x.grad.numpy(), "grad_x")
    s2 = s.detach().clone().requires_grad_(True)
    plx.LatticeFilterGeneral.apply(s2, x.detach(), dk).backward(gout)
    close(s2.grad.numpy(), "grad_src_only")


def test_shape_assert(cpu_method):
    dk = plx.DiscretizedKernelFN(plx.rbf, 1)
    with pytest.raises(AssertionError, match="Incompatible shapes"):
        plx.LatticeFilterGeneral.apply(torch.randn(4, 1), torch.randn(5, 2), dk)


def test_operator_surface(cpu_method):
    torch.manual_seed(0)
    k = plx.RBFLattice(order=1)
    assert isinstance(k, plx.LatticeAccelerated) and k.has_lengthscale
    assert abs(float(k.lengthscale) - 0.6931) < 1e-3            # softplus(0), GPyTorch's default
    x = torch.randn(40, 2)
    K = k(x, x)
    assert isinstance(K, plx.SquareLazyLattice)
    assert K.size() == torch.Size((40, 40)) and K.shape == (40, 40)
    assert K._transpose_nonbatch() is K
    assert torch.equal(K.diag(), torch.ones(40))
    assert torch.equal(k(x, x, diag=True), torch.ones(40))
    V = torch.randn(40, 3)
    want = oracle.filter(V.numpy(), (x / k.lengthscale).detach().numpy(), k.dkernel_fn.get_coeffs().numpy())
    np.testing.assert_allclose(K.matmul(V).detach().numpy(), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose((K @ V[:, 0]).detach().numpy(), want[:, 0], rtol=1e-5, atol=1e-6)

    # rectangular: pad + square filter over [xout; xin] + row slice (py:150-156)
    xs = torch.randn(15, 2)
    R = k(xs, x)
    assert isinstance(R, plx.RectangularLazyLattice)
    assert R.size() == torch.Size((15, 40))
    Rt = R._transpose_nonbatch()
    assert isinstance(Rt, plx.RectangularLazyLattice) and Rt.size() == torch.Size((40, 15))
    ell = k.lengthscale.detach()
    big_x = torch.cat([x / ell, xs / ell]).numpy()
    big_v = np.concatenate([V.numpy(), np.zeros((15, 3), np.float32)])
    want = oracle.filter(big_v, big_x, k.dkernel_fn.get_coeffs().numpy())[40:]
    np.testing.assert_allclose(R.matmul(V).detach().numpy(), want, rtol=1e-5, atol=1e-6)
    with pytest.raises(AssertionError, match="mismatched shapes"):
        R.matmul(torch.randn(7, 3))


def test_factories_defaults():
    assert plx.RBFLattice().dkernel_fn.order == 2                        # py:247
    assert plx.BilateralKernel(order=1).dkernel_fn.order == 1            # py:250-251
    m = plx.MaternLattice()
    assert m.dkernel_fn.order == 3                                       # py:253
    k = plx.MaternLattice(nu=2.5, order=1, ard_num_dims=5)
    assert k.lengthscale.shape[-1] == 5
    k.lengthscale = 2.0
    assert torch.allclose(k.lengthscale, torch.full((1, 5), 2.0), atol=1e-5)


def test_lengthscale_gradient_flows(cpu_method):
    k = plx.RBFLattice(order=1, ard_num_dims=2)
    x = torch.randn(30, 2, generator=torch.Generator().manual_seed(1))
    v = torch.randn(30, 1, generator=torch.Generator().manual_seed(2))
    (v * k(x, x).matmul(v)).sum().backward()
    g = k.raw_lengthscale.grad
    assert g is not None and g.shape == (1, 2) and torch.isfinite(g).all() and g.abs().sum() > 0


def test_no_cpu_fallback():
    """Without the test hook, CPU tensors are rejected loudly - never a silent CPU path."""
    plx.LatticeFilterGeneral.method = None
    dk = plx.DiscretizedKernelFN(plx.rbf, 1)
    with pytest.raises(ValueError, match="no CPU path"):
        plx.LatticeFilterGeneral.apply(torch.randn(4, 1), torch.randn(4, 2), dk)
    with pytest.raises(ValueError):
        plx.filter(torch.randn(4, 1), torch.randn(4, 2), torch.tensor([0.5, 1, 0.5]))


def test_c_abi_exports_every_declared_symbol():
    lib = _native.lib()
    declared = _native.declared_symbols()
    assert len(declared) >= 20 and "plx_filter" in declared and "plx_build" in declared
    for name in declared:
        assert hasattr(lib, name), f"libplx.so does not export {name}"
        assert name in _native._SIGNATURES, f"{name} has no ctypes signature"
    assert lib.plx_version().decode().startswith("libplx")
    assert lib.plx_strerror(3).decode().startswith("lattice coordinate")
    # argument validation happens before any GPU work
    h = ctypes.c_void_p()
    assert lib.plx_create(0, None) == 1
    assert lib.plx_num_vertices(None) == -1
    assert lib.plx_tune(b"no_such_key", 1) == 1 and b"no_such_key" in lib.plx_last_error()
    assert lib.plx_lattice_tune(None, b"xcd_remap", 1) == 1
    # the Lanczos step: its shape limits, and the argument checks that come before any launch
    assert lib.plx_lanczos_max_rows() == 256
    assert lib.plx_lanczos_work_floats(10_623) > 0 and lib.plx_lanczos_work_floats(2_097_152) > 0
    assert lib.plx_lanczos_work_floats(2_097_153) == -1 and lib.plx_lanczos_work_floats(0) == -1
    assert lib.plx_lanczos_step(None, 64, None, 64, 0, None, None, None, None) == 1 and b"NULL" in lib.plx_last_error()
    buf = (ctypes.c_float * 1024)()                                          # host memory: never reached by a launch below
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.plx_lanczos_step(p, 62, p, 62, 0, p, p, p, None) == 1 and b"multiple of 4" in lib.plx_last_error()
    assert lib.plx_lanczos_step(p, 64, p, 100, 0, p, p, p, None) == 1      # ld < n
    assert lib.plx_lanczos_step(p, 64, p, 64, 256, p, p, p, None) == 1     # more basis vectors than a step projects on
    # the diagnostic ablations are not part of the shipped library (libplx_diag.so only)
    if not os.environ.get("PLX_LIBRARY"):
        for key in (b"splat_ablate", b"blur_ablate", b"block_ablate"):
            assert lib.plx_tune(key, 1) == 1, key


def test_stale_library_is_refused(monkeypatch):
    """_native.lib() checks plx_version() against the ABI its ctypes signatures describe: a libplx.so left over from
    another release (an argument added in the middle of a signature) must fail at import, not with misaligned pointers."""
    lib = _native.lib()
    major, minor = (int(v) for v in lib.plx_version().decode().split()[1].split(".")[:2])
    assert (major, minor) == _native.ABI_VERSION
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "ABI_VERSION", (major, minor + 1))
    with pytest.raises(ImportError, match="rebuild"):
        _native.lib()
    monkeypatch.setattr(_native, "ABI_VERSION", (major, minor))
    assert _native.lib() is not None


def test_gp_compat_fallback_is_explicit():
    assert gp_compat.HAVE_GPYTORCH in (True, False)
    if not gp_compat.HAVE_GPYTORCH:
        assert plx.LatticeAccelerated.__mro__[1] is gp_compat._Kernel


def test_torch_extension_boundary_on_cpu():
    """The compiled PyTorch-ROCm extension (csrc/plx_torch.cpp) loads, exports the reference's one symbol and keeps the
    reference's input checks (cuda/permutohedral_cuda.cpp:3-5) as Python exceptions; compute needs a GPU (-m gpu)."""
    ext = plx.torch_ext.load()
    assert ext.version() == _native.lib().plx_version().decode()
    assert "filter(src" in ext.filter.__doc__ and hasattr(ext, "LatticeHandle")
    taps = torch.tensor([0.5, 1.0, 0.5])
    with pytest.raises(RuntimeError, match="must be a CUDA"):
        ext.filter(torch.randn(4, 1), torch.randn(4, 2), taps)


def test_lattice_cache_rebuilds_the_same_data_in_place_with_the_order_kept(monkeypatch):
    """Host logic of the point-order warm start (lattice_kernel._LatticeCache, no GPU): positions tagged as derived from the same
    data tensor (LatticeAccelerated.forward's position_hint) find the entry of that data and rebuild ITS lattice with
    reuse_order=True; other data, other taps or a data tensor changed in place take the ordinary path; the order is computed
    afresh after MAX_ORDER_AGE kept rebuilds; detach() / contiguous() copies carry the hint."""
    from simplex_gp_amd import lattice_kernel as lk

    class FakeLattice:
        made = 0

        def __init__(self, device=None):
            FakeLattice.made += 1
            self.builds, self.order_age, self.closed = [], -1, False

        def build(self, ref, taps, reuse_order=False):
            keep = reuse_order and self.order_age >= 0
            self.order_age = self.order_age + 1 if keep else 0
            self.builds.append(bool(keep))
            return self

        def close(self):
            self.closed = True

    monkeypatch.setattr(lk, "Lattice", FakeLattice)
    cache = lk._LatticeCache(capacity=4)
    x = torch.randn(16, 3)
    taps = np.array([0.3, 1.0, 0.3], np.float32)
    r0 = lk.position_hint(x / 0.7, x)
    lat0 = cache.get(r0, taps)
    assert cache.get(r0, taps) is lat0 and cache.hits == 1 and lat0.builds == [False]
    r1 = lk.position_hint(x / 0.8, x)
    assert cache.get(r1, taps) is lat0 and lat0.builds == [False, True] and cache.warm_rebuilds == 1 and len(cache._entries) == 1
    assert cache.get(lk.carry_hint(r1.detach(), r1), taps) is lat0 and cache.hits == 2          # same storage, same key: a hit
    other_taps = np.array([0.1, 0.5, 1.0, 0.5, 0.1], np.float32)
    lat1 = cache.get(lk.position_hint(x / 0.8, x), other_taps)                                   # other taps: a lattice of its own
    assert lat1 is not lat0 and lat1.builds == [False] and len(cache._entries) == 2
    x2 = x.clone()
    lat2 = cache.get(lk.position_hint(x2 / 0.8, x2), taps)                                       # other data: no warm start
    assert lat2 is not lat0 and lat2.builds == [False] and cache.warm_rebuilds == 1
    assert cache.get(x / 0.9, taps) not in (lat0, lat1, lat2)                                    # untagged positions: the ordinary path
    x.mul_(1.5)                                                                                  # the data changed in place: its old
    lat5 = cache.get(lk.position_hint(x / 0.8, x), taps)                                         # order says nothing about it
    assert lat5.builds[-1] is False and cache.warm_rebuilds == 1
    cache.clear()
    assert lat0.closed and lat1.closed
    # the order is refreshed after MAX_ORDER_AGE kept rebuilds, in place
    y = torch.randn(8, 2)
    lat = cache.get(lk.position_hint(y / 1.0, y), taps)
    for i in range(lk.MAX_ORDER_AGE + 3):
        assert cache.get(lk.position_hint(y / (1.0 + 0.01 * (i + 1)), y), taps) is lat
    assert lat.builds == [False] + [True] * lk.MAX_ORDER_AGE + [False, True, True] and len(cache._entries) == 1
    # the kernel's forward tags its scaled positions with the data they came from
    k = plx.RBFLattice(order=1)
    op = k(y, y)
    ((src, ver, shape),), scale = op.x._plx_positions_of
    assert src() is y and ver == y._version and shape == tuple(y.shape)
    assert scale[0]() is k.raw_lengthscale and scale[1] == k.raw_lengthscale._version
    # ... and with the state of the parameter their scale comes from: the same data under the SAME lengthscale, as a new
    # tensor (an evaluation, then the next training step), is served the lattice it has -- no build at all; a write to the
    # parameter (an optimiser step) makes it a re-scaled rebuild again
    cache.clear()
    lat = cache.get(k(y, y).x, taps)
    assert cache.get(k(y, y).x, taps) is lat and lat.builds == [False] and cache.same_positions == 1 and len(cache._entries) == 1
    with torch.no_grad():
        k.raw_lengthscale.add_(0.1)
    assert cache.get(k(y, y).x, taps) is lat and lat.builds == [False, True] and cache.same_positions == 1
    k2 = plx.RBFLattice(order=1)                                                                 # another kernel's lengthscale, equal
    assert cache.get(k2(y, y).x, taps) is lat and lat.builds == [False, True, True]               # or not: not the same state
    # the rectangular operator's stacked positions [xout; xin] carry both sources: the same two data tensors again (the
    # validation split, epoch after epoch) find their stacked lattice -- as it is under the same lengthscale, rebuilt in place
    # with the order kept under a moved one -- instead of piling up; another split is another lattice
    cache.clear()
    ys, yt = torch.randn(5, 2), torch.randn(6, 2)
    with torch.no_grad():
        R = k(ys, y)
        st = R._stacked_points()
        assert R._stacked_points() is st and len(st._plx_positions_of[0]) == 2 and st.shape == (13, 2)
        latr = cache.get(st, taps)
        assert cache.get(k(ys, y)._stacked_points(), taps) is latr and latr.builds == [False] and len(cache._entries) == 1
        k.raw_lengthscale.add_(0.05)
        assert cache.get(k(ys, y)._stacked_points(), taps) is latr and latr.builds == [False, True] and len(cache._entries) == 1
        assert cache.get(k(yt, y)._stacked_points(), taps) is not latr and len(cache._entries) == 2
