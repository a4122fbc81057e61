"""MI355X-native permutohedral-lattice K.v MVM behind the simplex-gp kernel API.

Public names mirror the reference package (gpytorch_lattice_kernel/__init__.py:1):
RBFLattice, MaternLattice (+ BilateralKernel), plus the native boundary
`filter(src, ref, coeffs)` and the staged `Lattice` handle.
"""
from .lattice import Lattice, filter  # noqa: F401
from .lattice_kernel import (  # noqa: F401
    BilateralKernel,
    LatticeAccelerated,
    LatticeFilterGeneral,
    MaternLattice,
    RBFLattice,
    RectangularLazyLattice,
    SquareLazyLattice,
    lattice_cache,
)
from .stencil import DiscretizedKernelFN, Matern, get_coeffs, matern, rbf  # noqa: F401
from . import distributed, solvers, torch_ext, training  # noqa: F401

__all__ = [
    "RBFLattice", "MaternLattice", "BilateralKernel", "LatticeAccelerated", "LatticeFilterGeneral",
    "SquareLazyLattice", "RectangularLazyLattice", "DiscretizedKernelFN", "get_coeffs", "rbf", "matern",
    "Matern", "Lattice", "filter", "lattice_cache",
]
