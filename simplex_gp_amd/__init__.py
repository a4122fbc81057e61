"""MI355X-native permutohedral-lattice K.v MVM behind the simplex-gp kernel API.

Public names mirror the reference package (gpytorch_lattice_kernel/__init__.py:1):
RBFLattice, MaternLattice (+ BilateralKernel), plus the native boundary
`filter(src, ref, coeffs)` and the staged `Lattice` handle.
"""
from .lattice import Lattice, filter  # noqa: F401

__all__ = ["Lattice", "filter"]
