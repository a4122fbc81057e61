"""The compiled PyTorch-ROCm extension `_plx_torch` (csrc/plx_torch.cpp): the reference's pybind11 boundary

    filter(src, ref, coeffs) -> Tensor        gpytorch_lattice_kernel/cuda/permutohedral_cuda.cpp:12-22

built ahead of time next to libplx.so (`make -C simplex_gp_amd/csrc`) instead of JIT-compiled at first use
(bilateral_kernel.py:62-74).  A reference maintainer swaps it in with one line:

    LatticeFilterGeneral.method = simplex_gp_amd.torch_ext.load().filter

The ctypes path (lattice.py) stays the default inside this package because it also reaches the staged entry
points; both sit on the same C ABI (include/plx.h).  No CPU fallback: a missing module raises.
"""
import importlib.util
import os

import torch  # noqa: F401  (libtorch must be loaded before the extension)

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "_plx_torch.so")
_mod = None


def load():
    """Import the prebuilt extension module (raises ImportError if it was not built)."""
    global _mod
    if _mod is None:
        if not os.path.exists(EXT_PATH):
            raise ImportError(f"{EXT_PATH} is missing: build it with `make -C simplex_gp_amd/csrc` "
                              "(or __graft_entry__.build()). There is no CPU fallback.")
        spec = importlib.util.spec_from_file_location("_plx_torch", EXT_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _mod = mod
    return _mod
