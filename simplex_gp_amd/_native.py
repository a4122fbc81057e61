"""ctypes binding of libplx.so (C ABI: include/plx.h).

There is no CPU fallback: if the library is missing or a call fails, this
module raises.  The product path never imports anything under oracle/.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLX_LIBRARY selects another build of the same C ABI -- libplx_diag.so (make -C simplex_gp_amd/csrc diag), which has the
# diagnostic ablation switches of tools/ablate_*.py compiled in.  Never a fallback: the named file must exist.
LIB_PATH = os.environ.get("PLX_LIBRARY") or os.path.join(_HERE, "libplx.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "plx.h")

PLX_OK = 0
ARRAY_KEYS, ARRAY_ENTRY_VERTEX, ARRAY_ENTRY_WEIGHT, ARRAY_NEIGHBORS = 0, 1, 2, 3
ARRAY_ROW_PTR, ARRAY_CSR_POINT, ARRAY_CSR_WEIGHT, ARRAY_POINT_PERM = 4, 5, 6, 7
MAX_DIM, MAX_ORDER = 32, 8
FACTOR_F32, FACTOR_F16 = 0, 1
ABI_VERSION = (0, 8)      # (major, minor) of plx_version() the signatures below belong to


class PlxError(RuntimeError):
    def __init__(self, code, where, detail):
        self.code = code
        super().__init__(f"{where}: {detail} (plx error {code})")


_lib = None
_vp, _i64, _i32, _f32p = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_float)

_SIGNATURES = {
    "plx_strerror": (ctypes.c_char_p, [_i32]),
    "plx_last_error": (ctypes.c_char_p, []),
    "plx_version": (ctypes.c_char_p, []),
    "plx_create": (_i32, [_i32, ctypes.POINTER(_vp)]),
    "plx_destroy": (None, [_vp]),
    "plx_build": (_i32, [_vp, _vp, _i64, _i32, _f32p, _i32, _i32, _i32, _vp]),
    "plx_build_local": (_i32, [_vp, _vp, _i64, _i32, _f32p, _i32, _vp]),
    "plx_key_words": (_i32, [_i32]),
    "plx_local_vertices": (_i64, [_vp]),
    "plx_copy_local_keys": (_i32, [_vp, _vp, _vp]),
    "plx_build_merge": (_i32, [_vp, _vp, ctypes.POINTER(_i64), _i32, _i32, _i64, _vp]),
    "plx_num_points": (_i64, [_vp]),
    "plx_num_owned": (_i64, [_vp]),
    "plx_num_vertices": (_i64, [_vp]),
    "plx_dim": (_i32, [_vp]),
    "plx_order": (_i32, [_vp]),
    "plx_device_bytes": (_i64, [_vp]),
    "plx_values_stride": (_i32, [_i32]),
    "plx_set_row_order": (_i32, [_vp, _i32]),
    "plx_splat": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "plx_splat_onehot": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "plx_filter_onehot": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp]),
    "plx_blur": (_i32, [_vp, _vp, _vp, _i32, ctypes.POINTER(_i32), _vp]),
    "plx_slice": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "plx_apply": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "plx_filter": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _f32p, _i32, _vp, _vp]),
    "plx_coldot": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "plx_coldot_work_floats": (_i64, [_i32]),
    "plx_apply_backward": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "plx_backward_stack": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "plx_backward_contract": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "plx_apply_affine": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "plx_affine_dot_work_floats": (_i64, [_vp, _i32]),
    "plx_apply_affine_dot": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "plx_set_reuse_order": (_i32, [_vp, _i32]),
    "plx_order_age": (_i32, [_vp]),
    "plx_affine_dot_tiles": (_i32, [_vp, _i32]),
    "plx_cg_fused_work_floats": (_i64, [_i32]),
    "plx_cg_step_update_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    "plx_cg_step_direction_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i32, _vp, _vp, _vp, _vp]),
    "plx_pcg_rz_partial_offset": (_i64, [_i32]),
    "plx_pcg_rz_partial_rows": (_i32, [_i64, _i32]),
    "plx_pcg_step_direction_fused": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i32, _vp, _vp, _vp, _vp,
                                            _vp]),
    "plx_lanczos_max_rows": (_i32, []),
    "plx_lanczos_work_floats": (_i64, [_i64]),
    "plx_lanczos_step": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "plx_cg_step_update": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "plx_cg_step_direction": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i32, _vp, _vp, _vp]),
    "plx_cg_update": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "plx_cg_direction": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "plx_pcg_work_floats": (_i64, [_i64, _i32, _i32]),
    "plx_pcg_project": (_i32, [_vp, _i32, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "plx_pcg_apply": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plx_pcg_factor_to_half": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "plx_pcg_step_direction": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i32, _vp, _vp, _vp]),
    "plx_pchol_work_bytes": (_i64, [_i64, _i32]),
    "plx_pchol_select": (_i32, [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp]),
    "plx_pchol_onehot": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp]),
    "plx_pchol_factor_batch": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i64, ctypes.c_float,
                                      _i32, _vp, _vp, _vp]),
    "plx_export": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "plx_export_bytes": (_i64, [_vp, _i32]),
    "plx_copy_point_perm": (_i32, [_vp, _vp, _vp]),
    "plx_tune": (_i32, [ctypes.c_char_p, _i32]),
    "plx_lattice_tune": (_i32, [_vp, ctypes.c_char_p, _i32]),
    "plx_last_kernels": (_i32, [_vp, ctypes.c_char_p, _i32]),
    "plx_block_rows": (_i64, [_vp]),
    "plx_prepare": (_i32, [_vp, _i32, _vp]),
    "plx_selftest_sort": (_i32, [_i64, _i32, _i32, ctypes.c_uint64, _vp, ctypes.POINTER(_i64)]),
    "plx_set_timing": (_i32, [_vp, _i32]),
    "plx_build_times": (_i32, [_vp, _f32p]),
    "plx_reference_growth_info": (_i32, [_vp, ctypes.POINTER(_i64)]),
    "plx_apply_times": (_i32, [_vp, _f32p, _i32, ctypes.POINTER(_i32)]),
}


def declared_symbols():
    """Function names declared in include/plx.h (used by the CPU-side ABI test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plx_[a-z_0-9]+)\s*\(", text)))


def lib():
    """Load libplx.so; raise if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C simplex_gp_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        # the argtypes below describe ONE version of the C ABI: a stale library would be called with misaligned
        # arguments and no diagnostic, so the (major, minor) of plx_version() must be the one this file was written for
        L.plx_version.restype = ctypes.c_char_p
        got = L.plx_version().decode()
        m = re.match(r"libplx (\d+)\.(\d+)\.", got)
        if m is None or (int(m.group(1)), int(m.group(2))) != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} reports '{got}', these bindings need ABI {ABI_VERSION[0]}.{ABI_VERSION[1]}.x: "
                              "rebuild it with `make -C simplex_gp_amd/csrc` (or __graft_entry__.build())")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, where):
    if rc != PLX_OK:
        L = lib()
        detail = L.plx_last_error().decode() or L.plx_strerror(rc).decode()
        raise PlxError(rc, where, detail)
