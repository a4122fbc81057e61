"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/lattice_oracle.c.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package (simplex_gp_amd).  numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "liblattice_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_i16p = ctypes.POINTER(ctypes.c_short)
_i8p = ctypes.POINTER(ctypes.c_byte)


def build():
    """Compile the C restatement (gcc, seconds)."""
    src = os.path.join(HERE, "lattice_oracle.c")
    if (not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.plxo_build.restype = ctypes.c_void_p
        L.plxo_build.argtypes = [_f32p, ctypes.c_long, ctypes.c_int, _f32p, ctypes.c_int]
        L.plxo_free.argtypes = [ctypes.c_void_p]
        L.plxo_num_vertices.restype = ctypes.c_long
        L.plxo_num_vertices.argtypes = [ctypes.c_void_p]
        L.plxo_grow_lookups.restype = ctypes.c_long
        L.plxo_grow_lookups.argtypes = [ctypes.c_void_p]
        for name, rt in [("plxo_keys", _i16p), ("plxo_entry_vertex", _i32p),
                         ("plxo_entry_weight", _f32p), ("plxo_greedy", _i16p),
                         ("plxo_rank", _i8p), ("plxo_scale", _f32p)]:
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.plxo_splat.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _f32p]
        L.plxo_neighbors.argtypes = [ctypes.c_void_p, ctypes.c_int, _i32p]
        L.plxo_blur.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, ctypes.c_int]
        L.plxo_slice.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _f32p]
        L.plxo_filter.restype = ctypes.c_int
        L.plxo_filter.argtypes = [_f32p, _f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                  _f32p, ctypes.c_int, _f32p, ctypes.POINTER(ctypes.c_long)]
        L.plxo_variance.restype = ctypes.c_float
        L.plxo_variance.argtypes = [_f32p, ctypes.c_int]
        L.plxo_scale_factors.argtypes = [ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        L.plxo_set_exact_mode.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def set_exact_mode(on):
    """True (default): reproduce the reference's stale-bucket-after-grow quirk
    (duplicate vertices) bit for bit.  False: duplicate-free lattice."""
    lib().plxo_set_exact_mode(1 if on else 0)


def variance(coeffs):
    c = _f32(coeffs)
    return float(lib().plxo_variance(_p(c), len(c)))


def scale_factors(d, coeffs):
    c = _f32(coeffs)
    sf = np.empty(d, np.float32)
    lib().plxo_scale_factors(d, _p(c), len(c), _p(sf))
    return sf


def filter(src, ref, coeffs, return_m=False):
    """Same contract as the reference's filter(src, ref, coeffs) (cpp:6-16)."""
    src, ref, coeffs = _f32(src), _f32(ref), _f32(coeffs)
    n, vd = src.shape
    assert ref.shape[0] == n
    d = ref.shape[1]
    out = np.empty((n, vd), np.float32)
    m = ctypes.c_long(0)
    rc = lib().plxo_filter(_p(src), _p(ref), n, d, vd, _p(coeffs), len(coeffs), _p(out),
                           ctypes.byref(m))
    if rc != 0:
        raise ValueError("plxo_filter: bad arguments")
    return (out, m.value) if return_m else out


class Lattice:
    """Staged access: structure, then splat / blur / slice separately."""

    def __init__(self, ref, coeffs):
        self.ref = _f32(ref)
        self.coeffs = _f32(coeffs)
        self.n, self.d = self.ref.shape
        self._h = lib().plxo_build(_p(self.ref), self.n, self.d, _p(self.coeffs), len(self.coeffs))
        self.m = lib().plxo_num_vertices(self._h)
        self.grow_lookups = lib().plxo_grow_lookups(self._h)

    def close(self):
        if self._h:
            lib().plxo_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _arr(self, fn, shape, dtype):
        ptr = fn(self._h)
        return np.ctypeslib.as_array(ptr, shape=shape).astype(dtype, copy=True)

    @property
    def keys(self):            # [m, d] int16, first-touch order
        return self._arr(lib().plxo_keys, (self.m, self.d), np.int16)

    @property
    def entry_vertex(self):    # [n, d+1] int32
        return self._arr(lib().plxo_entry_vertex, (self.n, self.d + 1), np.int32)

    @property
    def entry_weight(self):    # [n, d+1] float32
        return self._arr(lib().plxo_entry_weight, (self.n, self.d + 1), np.float32)

    @property
    def greedy(self):
        return self._arr(lib().plxo_greedy, (self.n, self.d + 1), np.int16)

    @property
    def rank(self):
        return self._arr(lib().plxo_rank, (self.n, self.d + 1), np.int8)

    @property
    def scale(self):
        return self._arr(lib().plxo_scale, (self.d,), np.float32)

    def neighbors(self):       # [d+1, 2r, m] int32, -1 = absent
        r = len(self.coeffs) // 2
        nbr = np.empty((self.d + 1, 2 * r, self.m), np.int32)
        lib().plxo_neighbors(self._h, len(self.coeffs), _p(nbr, _i32p))
        return nbr

    def splat(self, src):
        src = _f32(src)
        vd = src.shape[1]
        values = np.zeros((self.m, vd), np.float32)
        lib().plxo_splat(self._h, _p(src), vd, _p(values))
        return values

    def blur(self, values):
        values = _f32(values).copy()
        lib().plxo_blur(self._h, _p(values), values.shape[1], _p(self.coeffs), len(self.coeffs))
        return values

    def slice(self, values):
        values = _f32(values)
        out = np.empty((self.n, values.shape[1]), np.float32)
        lib().plxo_slice(self._h, _p(values), values.shape[1], _p(out))
        return out

    def filter(self, src):
        return self.slice(self.blur(self.splat(src)))
