"""Operator surface: the autograd op and the GPyTorch-facing kernel classes.

Mirrors gpytorch_lattice_kernel/bilateral_kernel.py (same names, constructor
arguments, shapes and error behaviour):
    LatticeFilterGeneral      py:59-124    forward = one filter, backward = one wider filter
    SquareLazyLattice         py:127-140   RectangularLazyLattice   py:142-160
    LatticeAccelerated        py:183-200
    RBFLattice / BilateralKernel / MaternLattice   py:247-254

What differs on purpose: the native call goes to libplx (HIP, gfx950) through
the C ABI, and the lattice built for a given (positions, taps) pair is kept and
reused by every later MVM on the same positions (all CG iterations call
_matmul with the same x / lengthscale tensor), instead of being rebuilt inside
every filter call (permutohedral.h:272).
"""
from collections import OrderedDict

import torch
from torch.autograd import Function

from .gp_compat import Kernel, LazyTensor
from .lattice import Lattice, _taps_array
from .stencil import DiscretizedKernelFN, Matern, rbf


# ---- warm start of the point order across re-scaled rebuilds -----------------------------------------------------------
# A training step divides the SAME data tensor by a lengthscale that moved a little (py:198-200) and builds a new lattice on
# the result.  The build's first stage orders the points along a space-filling curve -- a locality device only, nothing in
# the structure depends on it -- and that order survives a re-scaling (measured, N = 1e6, d = 8, tools/ab_reuse_order_r6.py:
# the order of l = 0.69 kept for l x 0.8 ... x 2: build -0.13 ... -0.17 ms of 0.5 ... 2.1, the MVMs as fast or faster).
# So a scaled tensor remembers which data tensor it came from (position_hint), and a cache miss whose hint matches an
# entry's rebuilds THAT entry's lattice in place with its point order kept (Lattice.build(reuse_order=True)).
MAX_ORDER_AGE = 8          # rebuilds before the order is computed afresh (Adam at lr 0.1 moves a lengthscale < 2.2x in 8 steps)


def position_hint(scaled, source, scale_of=None):
    """Mark `scaled` as positions derived from the data tensor `source` by a re-scaling (returns `scaled`).
    scale_of: the tensor the scale is a deterministic function of (the kernel's raw lengthscale PARAMETER; the lengthscale
    itself is a fresh softplus output on every access).  Two hints with the same source whose scale_of is the same tensor
    holding the same VALUES mark bit-identical positions: the cache then serves the lattice it has instead of rebuilding it
    (an evaluation followed by the next training step: the same hyper-parameters twice).  The values are compared on the
    device (d floats and one read-back, only when identity and version counter already agree): the version counter alone
    cannot vouch for them -- torch.optim.Adam(fused=True) writes parameters without moving it (measured on this image).
    The hint: ((weak reference, version, shape) per source tensor, scale key); stack_hint() joins the hints of stacked parts."""
    import weakref
    try:
        scale_key = None if scale_of is None else (weakref.ref(scale_of), scale_of._version)
        scaled._plx_positions_of = (((weakref.ref(source), source._version, tuple(source.shape)),), scale_key)
    except (AttributeError, TypeError):
        pass
    return scaled


def stack_hint(stacked, parts):
    """`stacked` is the concatenation of `parts` (the rectangular operator's [xout; xin], py:150-156): if every part says
    which data tensor it was scaled from, and by the same scale, the stack says so for all of them -- the stacked lattice of
    the SAME two data tensors under a moved lengthscale (the validation split, epoch after epoch) is then rebuilt in place
    with its point order kept instead of piling up in the cache as a new lattice per epoch."""
    hints = [getattr(p, "_plx_positions_of", None) for p in parts]
    if any(h is None for h in hints):
        return stacked
    scale = hints[0][1]
    for h in hints[1:]:
        if (h[1] is None) != (scale is None) or (scale is not None and (h[1][0]() is not scale[0]() or h[1][1] != scale[1])):
            return stacked
    try:
        stacked._plx_positions_of = (tuple(src for h in hints for src in h[0]), scale)
    except (AttributeError, TypeError):
        pass
    return stacked


def carry_hint(dst, src):
    """dst is src under another tensor object (detach / contiguous): keep the hint."""
    h = getattr(src, "_plx_positions_of", None)
    if h is not None and dst is not src:
        try:
            dst._plx_positions_of = h
        except (AttributeError, TypeError):
            pass
    return dst


def _same_hint(a, b):
    """Both hints name the same live data tensors, in the same states, and none has been written since."""
    if a is None or b is None or len(a[0]) != len(b[0]):
        return False
    for (ra, va, sa), (rb, vb, sb) in zip(a[0], b[0]):
        src = ra()
        if src is None or src is not rb() or va != vb or sa != sb or src._version != va:
            return False
    return True


def _scale_tensor(hint):
    """The live parameter a hint's scale comes from, or None."""
    k = hint[1] if hint is not None else None
    return None if k is None else k[0]()


def _same_scale(a, b, snapshot):
    """Both hints name the same live scale parameter, its version counter has not moved since either was made, and it
    still holds the values `snapshot` (taken when the lattice was built) -- the last by comparison on the device."""
    ka, kb = a[1], b[1]
    if ka is None or kb is None or snapshot is None:
        return False
    p = ka[0]()
    if p is None or p is not kb[0]() or not (ka[1] == kb[1] == p._version) or snapshot.shape != p.shape:
        return False
    return bool(torch.equal(snapshot, p.detach()))


def _snapshot_scale(hint):
    p = _scale_tensor(hint)
    return None if p is None else p.detach().clone()


class _LatticeCache:
    """Small LRU of built lattices keyed on the position tensor and the taps.

    An entry holds a reference to its position tensor, so the allocator cannot
    hand the same address to different data while the entry lives; together
    with tensor._version that makes (data_ptr, _version, shape, taps) a sound key
    for every write torch knows about.  Writes that bypass the version counter
    (`x.data.copy_()`, a custom kernel or a DLPack alias writing into the same
    storage) are NOT seen: call lattice_cache().clear() after such a write.
    """

    def __init__(self, capacity=4):
        self.capacity = capacity
        self._entries = OrderedDict()
        self.hits = 0
        self.misses = 0
        self.warm_rebuilds = 0             # misses served by rebuilding an entry of the same data in place (point order kept)
        self.same_positions = 0            # new position tensors recognised as the positions of an entry (same data, same scale state)

    @staticmethod
    def _key(ref, taps):
        return (ref.device.index, ref.data_ptr(), ref._version, tuple(ref.shape), taps.tobytes())

    def get(self, ref, coeffs):
        taps = _taps_array(coeffs)
        key = self._key(ref, taps)
        hit = self._entries.get(key)
        if hit is not None:
            self._entries.move_to_end(key)
            self.hits += 1
            return hit[0]
        self.misses += 1
        # the same data under a lengthscale that moved: rebuild that entry's lattice in place, point order kept
        hint = getattr(ref, "_plx_positions_of", None)
        if hint is not None:
            for k2, (lat2, ref2, snap2) in self._entries.items():
                hint2 = getattr(ref2, "_plx_positions_of", None)
                if (k2[0], k2[3], k2[4]) == (key[0], key[3], key[4]) and _same_hint(hint, hint2):
                    del self._entries[k2]
                    if _same_scale(hint, hint2, snap2):
                        # the same data divided by the same lengthscale, as a new tensor: the positions this lattice was
                        # built on.  The entry moves under the new tensor (the one the caller will come back with).
                        self._entries[key] = (lat2, ref, snap2)
                        self.misses -= 1
                        self.hits += 1
                        self.same_positions += 1
                        return lat2
                    keep = 0 <= lat2.order_age < MAX_ORDER_AGE      # (an order that old is computed afresh, in place all the same)
                    lat2.build(ref, taps, reuse_order=keep)
                    self._entries[key] = (lat2, ref, _snapshot_scale(hint))
                    self.warm_rebuilds += 1 if keep else 0
                    return lat2
        if len(self._entries) >= self.capacity:
            _, (old, _, _) = self._entries.popitem(last=False)
            lat = old                      # recycle the device buffers of the evicted lattice
        else:
            lat = Lattice(ref.device)
        lat.build(ref, taps)
        self._entries[key] = (lat, ref, _snapshot_scale(hint))
        return lat

    def clear(self):
        for lat, _, _ in self._entries.values():
            lat.close()
        self._entries.clear()


_cache = _LatticeCache()


def lattice_cache():
    return _cache


def cached_filter(src, ref, coeffs):
    """filter(src, ref, coeffs) with the lattice for (ref, coeffs) reused when
    the same position tensor comes back (the CG loop, the backward pass)."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda):
        raise ValueError("simplex_gp_amd has no CPU path: tensors must live on an MI355X (cuda) device")
    if src.dtype != torch.float32 or ref.dtype != torch.float32:
        raise TypeError(f"float32 only (got src {src.dtype}, ref {ref.dtype}); the reference CPU path is fp32 (h:277-278)")
    if not ref.is_cuda or ref.device != src.device:
        # checked before the cache is touched: a bad call must not evict a good lattice
        raise ValueError(f"src ({src.device}) and ref ({ref.device}) must live on the same MI355X (cuda) device")
    lat = _cache.get(ref if ref.is_contiguous() else carry_hint(ref.contiguous(), ref), coeffs)
    return lat.apply(src)


class LatticeFilterGeneral(Function):
    """out = K(reference) @ source, K the lattice approximation of the kernel
    whose taps come from `kernel_fn` (a DiscretizedKernelFN).

    `method` is the native filter used, as in the reference (py:60, py:94-95):
    None selects the HIP path; tests may install another callable with the
    reference's filter(src, ref, coeffs) signature.
    """

    method = None
    fused_backward = True      # False: position gradient through plx_backward_stack / filter / plx_backward_contract

    @staticmethod
    def _filter():
        return LatticeFilterGeneral.method if LatticeFilterGeneral.method is not None else cached_filter

    @staticmethod
    def forward(ctx, source, reference, kernel_fn):
        assert source.shape[0] == reference.shape[0], \
            "Incompatible shapes {}, and {}".format(source.shape, reference.shape)
        coeffs = kernel_fn.get_coeffs()
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(source, reference)
            ctx.coeffs = coeffs
            ctx.deriv_coeffs = kernel_fn.get_deriv_coeffs()
        return LatticeFilterGeneral._filter()(source, carry_hint(reference.contiguous(), reference), coeffs)

    @staticmethod
    def backward(ctx, grad_output):
        filt = LatticeFilterGeneral._filter()
        grad_source = grad_reference = None
        with torch.no_grad():
            src, ref = ctx.saved_tensors
            g = grad_output
            L = src.shape[-1]
            d = ref.shape[-1]
            if ctx.needs_input_grad[0] and not ctx.needs_input_grad[1]:
                # K is treated as symmetric (py:110-111)
                grad_source = filt(g.contiguous(), ref.contiguous(), ctx.coeffs)
            native = LatticeFilterGeneral.method is None and g.is_cuda and g.dim() == 2
            if ctx.needs_input_grad[1] and native and LatticeFilterGeneral.fused_backward and Lattice.backward_fusable(L, d):
                # the whole of py:113-123 in one native call: the stacked matrix is never stored (plx_apply_backward)
                rc_ = ref if ref.is_contiguous() else carry_hint(ref.contiguous(), ref)
                lat = _cache.get(rc_, ctx.deriv_coeffs)
                grad_reference, grad_source = lat.apply_backward(g, src, rc_, want_grad_src=ctx.needs_input_grad[0])
            elif ctx.needs_input_grad[1] and native:
                # same computation, the stack and the contraction each as one native pass (plx_backward_*)
                import ctypes
                from . import _native as nv
                n = src.shape[0]
                gc, sc, rc_ = g.contiguous(), src.contiguous(), ref.contiguous()
                stacked = torch.empty((n, 2 * L * (1 + d)), dtype=torch.float32, device=g.device)
                ptr = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
                stream = ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
                with torch.cuda.device(g.device):
                    nv.check(nv.lib().plx_backward_stack(ptr(gc), ptr(sc), ptr(rc_), n, L, d, ptr(stacked), stream),
                             "plx_backward_stack")
                filtered = filt(stacked, rc_, ctx.deriv_coeffs)
                del stacked
                grad_reference = torch.empty_like(rc_)
                with torch.cuda.device(g.device):
                    nv.check(nv.lib().plx_backward_contract(ptr(gc), ptr(sc), ptr(rc_), ptr(filtered), n, L, d,
                                                            ptr(grad_reference), stream), "plx_backward_contract")
                if ctx.needs_input_grad[0]:
                    grad_source = filtered[:, :L].contiguous()   # filtered with the derivative taps (py:123)
            elif ctx.needs_input_grad[1]:
                # one filter with the derivative taps over [g, g (x) x, src, src (x) x]   (py:113-119)
                gx = (g[..., None] * ref[..., None, :])          # n x L x d
                sx = (src[..., None] * ref[..., None, :])
                stacked = torch.cat([g, gx.reshape(gx.shape[:-2] + (L * d,)),
                                     src, sx.reshape(sx.shape[:-2] + (L * d,))], dim=-1)
                filtered = filt(stacked.contiguous(), ref.contiguous(), ctx.deriv_coeffs)
                wg, wgx, ws, wsx = torch.split(filtered, [L, L * d, L, L * d], dim=-1)
                wgx = wgx.reshape(-1, L, d)
                wsx = wsx.reshape(-1, L, d)
                # py:122
                grad_reference = -2 * (sx * wg[..., None] - src[..., None] * wgx
                                       + gx * ws[..., None] - g[..., None] * wsx).sum(-2)
                if ctx.needs_input_grad[0]:
                    grad_source = wg       # filtered with the derivative taps, as the reference does (py:123)
        return grad_source, grad_reference, None


def _lattice_matvec(rhs, positions, dkernel):
    """K(positions) @ rhs through the autograd op; the one place the operator classes reach the native filter."""
    return LatticeFilterGeneral.apply(rhs, positions, dkernel)


class SquareLazyLattice(LazyTensor):
    """K(x, x), known through its action only (py:127-140): symmetric by declaration (its transpose is itself,
    py:137-138) and with a unit diagonal by declaration (py:139-140), whatever the lattice actually computes."""

    def __init__(self, x, dkernel=None):
        super().__init__(x, dkernel=dkernel)
        self.x, self.dkernel = x, dkernel

    def _size(self):
        n = self.x.shape[-2]
        return torch.Size((n, n))

    def _matmul(self, V):
        return _lattice_matvec(V, self.x, self.dkernel)

    def _transpose_nonbatch(self):
        return self

    def diag(self):
        return self.x.new_ones(self.x.shape[:-1])


class RectangularLazyLattice(LazyTensor):
    """K(xin, xout) for two different point sets (prediction), py:142-160: the right-hand side lives on `xout`; it is
    extended by zeros over `xin`, ONE square filter runs over the stacked points [xout; xin], and the rows that belong to
    `xin` are returned."""

    def __init__(self, xin, xout, dkernel=None):
        super().__init__(xin, xout, dkernel=dkernel)
        self.xin, self.xout, self.dkernel = xin, xout, dkernel

    def _size(self):
        return torch.Size((*self.xin.shape[:-1], self.xout.shape[-2]))

    def _matmul(self, V):
        n_out, n_in = self.xout.shape[-2], self.xin.shape[-2]
        assert V.shape[-2] == n_out, f"mismatched shapes? {V.shape, self.xout.shape}"
        stacked_rhs = torch.nn.functional.pad(V, (0, 0, 0, n_in))          # zero rows for the points of xin
        return _lattice_matvec(stacked_rhs, self._stacked_points(), self.dkernel)[..., n_out:, :]

    def _stacked_points(self):
        """[xout; xin].  Without a gradient to carry, the stacked tensor is made once per operator (and per state of its
        two inputs): a fresh concatenation on every product is a fresh lattice-cache key, i.e. one lattice build per product
        (measured, N = 1e6 + 2.5e5 points: 1.5 of the 9.8 ms of every K(x*, x) @ V on the same operator)."""
        if torch.is_grad_enabled() and (self.xin.requires_grad or self.xout.requires_grad):
            return torch.cat((self.xout, self.xin), dim=-2)
        key = (self.xin._version, self.xout._version)
        hit = self.__dict__.get("_stacked")
        if hit is None or hit[0] != key:
            hit = (key, stack_hint(torch.cat((self.xout.detach(), self.xin.detach()), dim=-2), (self.xout, self.xin)))
            self.__dict__["_stacked"] = hit
        return hit[1]

    def _transpose_nonbatch(self):
        return type(self)(self.xout, self.xin, self.dkernel)


class LatticeAccelerated(Kernel):
    """A stationary kernel, given as a differentiable profile in the squared distance, evaluated through the
    permutohedral lattice (py:183-200).  ARD lengthscales divide the inputs before they reach the lattice (py:198-200);
    the diagonal is reported as ones (py:193-195)."""

    has_lengthscale = True

    def __init__(self, kernel_fn, *args, order=2, **kwargs):
        super().__init__(*args, **kwargs)
        self.dkernel_fn = DiscretizedKernelFN(kernel_fn, order)

    @staticmethod
    def _same_points(a, b):
        """The reference decides square vs rectangular by comparing the two tensors element by element on every call
        (py:197, one device synchronisation); identical objects / storage are recognised without that."""
        if a is b:
            return True
        if a.shape != b.shape:
            return False
        return a.data_ptr() == b.data_ptr() or bool(torch.equal(a, b))

    def forward(self, x1, x2, diag=False, **params):
        if diag:
            return x1.new_ones(x1.shape[:-1])
        scaled1 = position_hint(x1.div(self.lengthscale), x1, scale_of=getattr(self, "raw_lengthscale", None))
        if self._same_points(x1, x2):
            return SquareLazyLattice(scaled1, self.dkernel_fn)
        scaled2 = position_hint(x2.div(self.lengthscale), x2, scale_of=getattr(self, "raw_lengthscale", None))
        return RectangularLazyLattice(scaled1, scaled2, self.dkernel_fn)


def _lattice_kernel_factory(profile, default_order):
    def make(*args, order=default_order, **kwargs):
        return LatticeAccelerated(profile, *args, order=order, **kwargs)
    return make


RBFLattice = _lattice_kernel_factory(rbf, 2)                    # py:247-248
RBFLattice.__name__ = "RBFLattice"
BilateralKernel = RBFLattice                                    # py:250-251


def MaternLattice(*args, nu=1.5, order=3, **kwargs):            # py:253-254
    return LatticeAccelerated(lambda d2: Matern.apply(d2, nu), *args, order=order, **kwargs)
