"""Device lattice handle + the reference's `filter(src, ref, coeffs)` boundary.

`filter` is the drop-in for the one native symbol of the reference
(gpytorch_lattice_kernel/cpp/lattice.cpp:6-16, cuda/permutohedral_cuda.cpp:12-22):
same argument order, shapes, dtype and return value.  `Lattice` is the staged
form (build once, apply many times) that the CG loop and the backward pass use.

torch is plumbing here: device memory, the current stream, nothing else.
"""
import ctypes

import numpy as np
import torch

from . import _native as nv


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _taps_array(coeffs):
    if isinstance(coeffs, torch.Tensor):
        coeffs = coeffs.detach().to("cpu", torch.float32).numpy()
    taps = np.ascontiguousarray(coeffs, dtype=np.float32).reshape(-1)
    if taps.size % 2 != 1:
        raise ValueError(f"coeffs must have an odd number of taps, got {taps.size}")
    return taps


def _check_f32_cuda(t, name, ndim=2):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA (HIP) tensor; this build has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.dim() != ndim:
        raise ValueError(f"{name} must be {ndim}-D, got shape {tuple(t.shape)}")


class Lattice:
    """One permutohedral lattice on one GPU.

    build(ref, coeffs) fixes the structure (vertices, barycentric weights,
    blur neighbour table, splat CSR); apply(src) is one K.v MVM on it.
    Not thread-safe (ping-pong workspace), same as documented in plx.h.
    """

    def __init__(self, device=None):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self._h = ctypes.c_void_p()
        nv.check(nv.lib().plx_create(self.device.index, ctypes.byref(self._h)), "plx_create")
        self._ref = None          # keeps the positions alive while kernels may still read them
        self.taps = None
        self._perm_cache = None
        self._own_begin = 0
        self.lattice_rows = False
        self._dot_work = None
        self.build_id = 0         # counts the builds of this object (holders of per-build data compare it)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            nv.lib().plx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- structure --------------------------------------------------------
    def build(self, ref, coeffs, shard=None, reuse_order=False):
        """shard = (index, count): this process splats / slices block `index` of the
        `count` contiguous near-equal row blocks (distributed.shard_bounds); None = all rows.
        reuse_order=True: `ref` is the previous build's positions re-scaled a little (a lengthscale that moved): keep the
        lattice order of the points (plx_set_reuse_order: the order passes of the build are skipped; the result is the
        cold build's up to the order of the fp32 sums inside a vertex row).  Ignored when no order of that shape exists."""
        _check_f32_cuda(ref, "ref")
        ref = ref.contiguous()
        taps = _taps_array(coeffs)
        n, d = ref.shape
        index, count = (0, 1) if shard is None else shard
        if reuse_order:
            nv.check(nv.lib().plx_set_reuse_order(self._h, 1), "plx_set_reuse_order")
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_build(self._h, ctypes.c_void_p(ref.data_ptr()), n, d,
                                    taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), taps.size,
                                    index, count, _stream_ptr(self.device))
        nv.check(rc, "plx_build")
        self.build_id += 1
        self._ref = ref
        self.taps = taps
        self._perm_cache = None
        base, extra = divmod(n, count)
        self._own_begin = index * base + min(index, extra)
        return self

    def filter_once(self, src, ref, coeffs, out=None):
        """plx_filter on this lattice's buffers: build for `ref`, one MVM of `src`, in one native call.  The build knows
        the lattice serves a single MVM and leaves out what only pays back over several (vertex renumbering, axis-pair
        tables); the lattice stays usable afterwards like one built by build()."""
        _check_f32_cuda(ref, "ref")
        ref = ref.contiguous()
        taps = _taps_array(coeffs)
        n, d = ref.shape
        src = self._src(src, n)
        vd = src.shape[1]
        if out is None:
            out = torch.empty((n, vd), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_filter(self._h, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(ref.data_ptr()), n, d, vd,
                                     taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), taps.size,
                                     ctypes.c_void_p(out.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_filter")
        self.build_id += 1
        self._ref, self.taps, self._perm_cache, self._own_begin = ref, taps, None, 0
        return out

    # -- sharded build: local stage / key exchange / merge ---------------------
    def build_local(self, ref_local, coeffs):
        """Stage 1 of a sharded build: structure of this rank's own rows only.
        Returns the rank's vertex keys as an int32 tensor [m_local, key_words]."""
        _check_f32_cuda(ref_local, "ref_local")
        ref_local = ref_local.contiguous()
        taps = _taps_array(coeffs)
        n, d = ref_local.shape
        L = nv.lib()
        with torch.cuda.device(self.device):
            rc = L.plx_build_local(self._h, ctypes.c_void_p(ref_local.data_ptr()), n, d,
                                   taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), taps.size,
                                   _stream_ptr(self.device))
        nv.check(rc, "plx_build_local")
        self.build_id += 1
        self._ref, self.taps, self._perm_cache, self._own_begin = ref_local, taps, None, 0
        keys = torch.empty((int(L.plx_local_vertices(self._h)), int(L.plx_key_words(d))), dtype=torch.int32,
                           device=self.device)
        with torch.cuda.device(self.device):
            rc = L.plx_copy_local_keys(self._h, ctypes.c_void_p(keys.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_copy_local_keys")
        return keys

    def build_merge(self, all_keys, counts, rank, total_points=0):
        """Stage 2: all ranks' keys concatenated in rank order ([sum(counts), key_words] int32); total_points = rows of
        all ranks together (feeds the vertex-numbering choice only; 0 = unknown)."""
        all_keys = all_keys.contiguous()
        assert all_keys.is_cuda and all_keys.dtype == torch.int32 and all_keys.shape[0] == sum(counts)
        arr = (ctypes.c_int64 * len(counts))(*[int(c) for c in counts])
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_build_merge(self._h, ctypes.c_void_p(all_keys.data_ptr()), arr, len(counts), rank,
                                          int(total_points), _stream_ptr(self.device))
        nv.check(rc, "plx_build_merge")
        return self

    @property
    def order_age(self):
        """Builds since the point order was computed from the positions themselves (0: by the last build; -1: none)."""
        return int(nv.lib().plx_order_age(self._h))

    @property
    def n(self):
        return int(nv.lib().plx_num_points(self._h))

    @property
    def n_owned(self):
        return int(nv.lib().plx_num_owned(self._h))

    @property
    def m(self):
        return int(nv.lib().plx_num_vertices(self._h))

    @property
    def d(self):
        return int(nv.lib().plx_dim(self._h))

    @property
    def order(self):
        return int(nv.lib().plx_order(self._h))

    @property
    def device_bytes(self):
        return int(nv.lib().plx_device_bytes(self._h))

    # -- row order ----------------------------------------------------------
    def shard_perm(self):
        """int64 device tensor: caller row (within this shard) of the i-th row in lattice order."""
        if self._perm_cache is None:
            raw = torch.empty(self.n, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                rc = nv.lib().plx_copy_point_perm(self._h, ctypes.c_void_p(raw.data_ptr()), _stream_ptr(self.device))
            nv.check(rc, "plx_copy_point_perm")
            begin = self._own_begin        # rows of this shard occupy the same index range in both orders
            self._perm_cache = raw[begin:begin + self.n_owned].to(torch.int64) - begin
        return self._perm_cache

    def set_lattice_row_order(self, on=True):
        """on: splat/slice/apply take and return rows in lattice order (no per-MVM permutation)."""
        nv.check(nv.lib().plx_set_row_order(self._h, 1 if on else 0), "plx_set_row_order")
        self.lattice_rows = bool(on)

    def to_lattice_order(self, t):
        return t.index_select(0, self.shard_perm())

    def from_lattice_order(self, t):
        out = torch.empty_like(t)
        out.index_copy_(0, self.shard_perm(), t)
        return out

    def set_timing(self, on=True):
        nv.check(nv.lib().plx_set_timing(self._h, 1 if on else 0), "plx_set_timing")

    def reference_growth_info(self):
        """What the replay of the reference CPU path's table-growth quirk found for this build (plx_tune
        "reference_growth"): {"replayed", "m_reference", "dropped", "invisible", "blur_miss", "inexact"}."""
        buf = (ctypes.c_int64 * 6)()
        nv.check(nv.lib().plx_reference_growth_info(self._h, buf), "plx_reference_growth_info")
        return {"replayed": bool(buf[0]), "m_reference": int(buf[1]), "dropped": int(buf[2]), "invisible": int(buf[3]),
                "blur_miss": bool(buf[4]), "inexact": bool(buf[5])}

    def build_times_ms(self):
        buf = (ctypes.c_float * 6)()
        nv.check(nv.lib().plx_build_times(self._h, buf), "plx_build_times")
        return dict(zip(("embed", "insert", "number", "ids", "neighbours", "csr"), list(buf)))

    def stage_kernels(self, vd=None):
        """Kernels launched by the last splat / blur / slice on this lattice: {"splat": [...], "blur_axis": [...],
        "slice": [...]} (names as rocprofv3 shows them, without template arguments)."""
        buf = ctypes.create_string_buffer(512)
        nv.check(nv.lib().plx_last_kernels(self._h, buf, 512), "plx_last_kernels")
        out = {}
        for part in buf.value.decode().split(";"):
            k, _, names = part.partition("=")
            out[k] = [x for x in names.split("+") if x]
        return out

    def prepare(self, vd=1):
        """Build now every table an MVM with vd columns will read (otherwise the first such MVM builds them)."""
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_prepare(self._h, int(vd), _stream_ptr(self.device))
        nv.check(rc, "plx_prepare")
        return self

    def tune(self, key, value):
        """Switch a kernel variant for THIS lattice only, from its next call until its next build (plx_lattice_tune;
        the process defaults -- plx_tune -- are what every build starts from)."""
        nv.check(nv.lib().plx_lattice_tune(self._h, key.encode(), int(value)), "plx_lattice_tune")
        return self

    @property
    def block_rows(self):
        """Rows of the block tables (0: not built yet -- see prepare() -- or the lattice uses the vertex-sorted CSR path)."""
        return int(nv.lib().plx_block_rows(self._h))

    def apply_times_ms(self):
        """Stage times of the last apply(): dict(splat, blur, slice) in ms (blur = all d+1 launches)."""
        cap = 8
        buf = (ctypes.c_float * cap)()
        cnt = ctypes.c_int(0)
        nv.check(nv.lib().plx_apply_times(self._h, buf, cap, ctypes.byref(cnt)), "plx_apply_times")
        t = list(buf)[:cnt.value]
        if len(t) != 3:
            return None
        return {"splat": t[0], "blur": t[1], "slice": t[2]}

    # -- stages -----------------------------------------------------------
    def _src(self, src, rows):
        _check_f32_cuda(src, "src")
        if src.shape[0] != rows:
            raise ValueError(f"Incompatible shapes {tuple(src.shape)}, expected {rows} rows")
        return src.contiguous()

    @staticmethod
    def values_stride(vd):
        """Floats per vertex row of a values buffer (vd rounded up to 4 when vd > 1)."""
        return int(nv.lib().plx_values_stride(vd))

    def new_values(self, vd):
        """Vertex accumulator [m, values_stride(vd)]; columns >= vd are zero padding."""
        return torch.empty((self.m, self.values_stride(vd)), dtype=torch.float32, device=self.device)

    def splat(self, src, values=None):
        src = self._src(src, self.n_owned)
        vd = src.shape[1]
        if values is None:
            values = self.new_values(vd)
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_splat(self._h, ctypes.c_void_p(src.data_ptr()), vd,
                                    ctypes.c_void_p(values.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_splat")
        return values

    def splat_onehot(self, points, nb, values, vd=None):
        """values[:, b] = S^T e_p for p = points[b] (device int32, lattice-order point indices), b < nb; zero elsewhere."""
        vd = values.shape[1] if vd is None else vd
        assert values.shape[1] == self.values_stride(vd) and values.is_contiguous() and points.dtype == torch.int32
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_splat_onehot(self._h, ctypes.c_void_p(points.data_ptr()), int(nb), vd,
                                           ctypes.c_void_p(values.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_splat_onehot")
        return values

    def filter_onehot(self, points, nb, values, scratch, out, vd=None, sparse=True, frontier=None):
        """out[:, b] = K e_p for p = points[b] (device int32, lattice-order point indices), b < nb <= 16: splat_onehot +
        blur + slice in one call; sparse=True runs them on the frontier of the columns' non-zero vertex rows
        (plx_filter_onehot).  values / scratch: two [m, values_stride(vd)] buffers (clobbered); out: [n, vd], rows in
        the order set_lattice_row_order() says; frontier (optional device int32 [1]) receives the vertex rows the last
        blur axis worked on."""
        vd = out.shape[1] if vd is None else vd
        _check_f32_cuda(values, "values"); _check_f32_cuda(scratch, "scratch"); _check_f32_cuda(out, "out")
        assert values.shape[1] == self.values_stride(vd) and values.is_contiguous() and scratch.shape == values.shape
        assert scratch.is_contiguous() and values.shape[0] >= self.m and points.dtype == torch.int32 and points.numel() >= nb
        assert out.is_contiguous() and out.shape == (self.n_owned, vd)
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_filter_onehot(self._h, ctypes.c_void_p(points.data_ptr()), int(nb), vd,
                                            ctypes.c_void_p(values.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                                            ctypes.c_void_p(out.data_ptr()), 1 if sparse else 0,
                                            ctypes.c_void_p(frontier.data_ptr()) if frontier is not None else None,
                                            _stream_ptr(self.device))
        nv.check(rc, "plx_filter_onehot")
        return out

    def blur(self, values, scratch=None, vd=None):
        """Returns the tensor holding the blurred values (either `values` or `scratch`).
        `values` is [m, values_stride(vd)]; vd defaults to its width."""
        _check_f32_cuda(values, "values")
        vd = values.shape[1] if vd is None else vd
        assert values.shape[1] == self.values_stride(vd) and values.is_contiguous()
        if values.shape[0] < self.m:
            raise ValueError(f"values has {values.shape[0]} rows, the lattice {self.m} vertices")
        if scratch is None:
            scratch = torch.empty_like(values)
        elif scratch.numel() < values.numel() or not scratch.is_contiguous():
            raise ValueError("scratch must be a contiguous buffer at least as large as values")
        flag = ctypes.c_int(0)
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_blur(self._h, ctypes.c_void_p(values.data_ptr()),
                                   ctypes.c_void_p(scratch.data_ptr()), vd, ctypes.byref(flag),
                                   _stream_ptr(self.device))
        nv.check(rc, "plx_blur")
        return scratch if flag.value else values

    def slice(self, values, out=None, vd=None):
        _check_f32_cuda(values, "values")
        vd = values.shape[1] if vd is None else vd
        assert values.shape[1] == self.values_stride(vd) and values.is_contiguous()
        if out is None:
            out = torch.empty((self.n_owned, vd), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_slice(self._h, ctypes.c_void_p(values.data_ptr()), vd,
                                    ctypes.c_void_p(out.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_slice")
        return out

    def apply(self, src, out=None):
        """One MVM: out = slice(blur(splat(src))) on the lattice's own workspace."""
        src = self._src(src, self.n_owned)
        vd = src.shape[1]
        if out is None:
            out = torch.empty((self.n_owned, vd), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_apply(self._h, ctypes.c_void_p(src.data_ptr()), vd,
                                    ctypes.c_void_p(out.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_apply")
        return out

    def apply_affine(self, src, scale_shift, out=None, want_dot=False):
        """out = a * K src + b * src with (a, b) = scale_shift (a 2-element float32 tensor on the device, read
        there: no host synchronisation on hyper-parameters).  want_dot: also return the column-wise <src, out>
        (the p^T A p of a CG iteration), formed inside the slice kernel; needs 2..256 columns.  want_dot="partial":
        return (out, work, tiles) with the per-tile partial sums of those dots left un-reduced in `work` (the fused CG
        step adds them up itself); `work` is the lattice's own buffer, valid until the next such call."""
        src = self._src(src, self.n_owned)
        _check_f32_cuda(scale_shift, "scale_shift", ndim=1)
        assert scale_shift.numel() == 2 and scale_shift.is_contiguous()
        vd = src.shape[1]
        if out is None:
            out = torch.empty((self.n_owned, vd), dtype=torch.float32, device=self.device)
        if want_dot:
            L = nv.lib()
            need = int(L.plx_affine_dot_work_floats(self._h, vd))
            if need < 0:
                raise ValueError(f"apply_affine(want_dot=True) needs 2..256 columns on a built lattice, got {vd}")
            if self._dot_work is None or self._dot_work.numel() < need:
                self._dot_work = torch.empty(need, dtype=torch.float32, device=self.device)
            if want_dot == "partial":
                # the slice kernel's per-tile partial sums stay in the work buffer: (out, partials, tiles) for
                # plx_cg_step_update_fused, which adds them up itself (no stand-alone reduction launch)
                with torch.cuda.device(self.device):
                    rc = L.plx_apply_affine_dot(self._h, ctypes.c_void_p(src.data_ptr()), vd, ctypes.c_void_p(out.data_ptr()),
                                                ctypes.c_void_p(scale_shift.data_ptr()), None,
                                                ctypes.c_void_p(self._dot_work.data_ptr()), _stream_ptr(self.device))
                nv.check(rc, "plx_apply_affine_dot")
                return out, self._dot_work, int(L.plx_affine_dot_tiles(self._h, vd))
            dot = torch.empty(self.values_stride(vd), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.plx_apply_affine_dot(self._h, ctypes.c_void_p(src.data_ptr()), vd, ctypes.c_void_p(out.data_ptr()),
                                            ctypes.c_void_p(scale_shift.data_ptr()), ctypes.c_void_p(dot.data_ptr()),
                                            ctypes.c_void_p(self._dot_work.data_ptr()), _stream_ptr(self.device))
            nv.check(rc, "plx_apply_affine_dot")
            return out, dot[:vd]
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_apply_affine(self._h, ctypes.c_void_p(src.data_ptr()), vd, ctypes.c_void_p(out.data_ptr()),
                                           ctypes.c_void_p(scale_shift.data_ptr()), _stream_ptr(self.device))
        nv.check(rc, "plx_apply_affine")
        return out

    @staticmethod
    def backward_fusable(nrhs, d):
        """True when plx_apply_backward covers this shape: 125..512 columns and 2*nrhs + d <= 62."""
        return 32 <= (2 * nrhs * (1 + d) + 3) // 4 <= 128 and 2 * nrhs + d <= 62

    def apply_backward(self, grad_out, src, ref, want_grad_src=True):
        """Position gradient of out = K(ref) src for a lattice built on `ref` with the DERIVATIVE taps
        (bilateral_kernel.py:113-123), fused: returns (grad_ref [n, d], grad_src [n, nrhs] or None)."""
        g = self._src(grad_out, self.n_owned)
        src = self._src(src, self.n_owned)
        ref = self._src(ref, self.n_owned)
        if g.shape != src.shape or ref.shape[1] != self.d:
            raise ValueError(f"Incompatible shapes {tuple(g.shape)}, {tuple(src.shape)}, {tuple(ref.shape)}")
        grad_ref = torch.empty_like(ref)
        grad_src = torch.empty_like(src) if want_grad_src else None
        with torch.cuda.device(self.device):
            rc = nv.lib().plx_apply_backward(self._h, ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(src.data_ptr()),
                                             ctypes.c_void_p(ref.data_ptr()), src.shape[1],
                                             ctypes.c_void_p(grad_ref.data_ptr()),
                                             ctypes.c_void_p(grad_src.data_ptr()) if want_grad_src else None,
                                             _stream_ptr(self.device))
        nv.check(rc, "plx_apply_backward")
        return grad_ref, grad_src

    # -- introspection (parity tests) --------------------------------------
    def export(self, which):
        L = nv.lib()
        nbytes = int(L.plx_export_bytes(self._h, which))
        if nbytes < 0:
            raise ValueError(f"unknown array {which}")
        n, m, d, r = self.n, self.m, self.d, self.order
        dtype, shape = {
            nv.ARRAY_KEYS: (np.int16, (m, d)),
            nv.ARRAY_ENTRY_VERTEX: (np.int32, (d + 1, n)),
            nv.ARRAY_ENTRY_WEIGHT: (np.float32, (d + 1, n)),
            nv.ARRAY_NEIGHBORS: (np.int32, (d + 1, 2 * r, m)),
            nv.ARRAY_ROW_PTR: (np.int32, (m + 1,)),
            nv.ARRAY_CSR_POINT: (np.int32, (self.n_owned * (d + 1),)),
            nv.ARRAY_CSR_WEIGHT: (np.float32, (self.n_owned * (d + 1),)),
            nv.ARRAY_POINT_PERM: (np.uint32, (n,)),
        }[which]
        out = np.empty(shape, dtype)
        assert out.nbytes == nbytes, (out.nbytes, nbytes)
        with torch.cuda.device(self.device):
            rc = L.plx_export(self._h, which, out.ctypes.data_as(ctypes.c_void_p), nbytes,
                              _stream_ptr(self.device))
        nv.check(rc, "plx_export")
        return out


_scratch = {}


def _scratch_lattice(device):
    key = device.index
    lat = _scratch.get(key)
    if lat is None:
        lat = _scratch[key] = Lattice(device)
    return lat


def filter(src, ref, coeffs):
    """filter(src[N,vd], ref[N,d], coeffs[R]) -> out[N,vd]   (reference boundary).

    Builds a fresh lattice for `ref` on every call, exactly like the reference
    (permutohedral.h:272); only the device buffers are recycled between calls.
    """
    _check_f32_cuda(src, "src")
    _check_f32_cuda(ref, "ref")
    if src.shape[0] != ref.shape[0]:
        raise ValueError("Incompatible shapes {}, and {}".format(tuple(src.shape), tuple(ref.shape)))
    if src.device != ref.device:
        raise ValueError("src and ref must be on the same device")
    return _scratch_lattice(src.device).filter_once(src, ref, coeffs)
