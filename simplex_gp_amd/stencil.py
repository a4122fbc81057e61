"""Stencil taps for a stationary kernel profile, and the profiles themselves.

Host-side, runs once per kernel object.  Mirrors (same names, argument meaning
and numerics) gpytorch_lattice_kernel/bilateral_kernel.py:
    get_coeffs        py:14-28     coverage_diff   py:30-39
    binary_search     py:41-56     DiscretizedKernelFN  py:162-181
    rbf               py:202-203   Matern / matern      py:207-245
The expected tap vectors are pinned by tests/golden/host_side.npz.
"""
import logging
import math

import numpy as np
import torch
import torch.nn as nn

_log = logging.getLogger(__name__)

_GRID_POINTS = 10 ** 4     # samples of the profile on [-30, 30]          (py:19-20)
_GRID_HALF_WIDTH = 30.0
_SPACING_BOUNDS = (0.1, 9.0)   # bisection bracket for the tap spacing     (py:26)
_SPACING_TOL = 1e-4


def coverage_diff(spacing, order, x, w, fn_values, fft_values):
    """Spatial coverage of R = 2*order+1 taps at `spacing` minus the spectral
    coverage below the Nyquist frequency pi/spacing (py:30-39)."""
    taps = 2 * order + 1
    half_extent = spacing * taps / 2
    nyquist = np.pi / spacing
    in_space = (x >= -half_extent) & (x <= half_extent)
    in_band = (w >= -nyquist) & (w <= nyquist)
    spatial = fn_values[in_space].sum() / fn_values.sum()
    spectral = fft_values[in_band].sum() / fft_values.sum()
    _log.debug("coverage: spatial %.3f spectral %.3f at spacing %.4f", spatial, spectral, spacing)
    return spatial - spectral


def binary_search(target, bounds, fn, eps=1e-2):
    """Bisection for fn(x) = target on a bracket, fn increasing (py:41-56)."""
    lo, hi = bounds
    for _ in range(501):
        if hi - lo <= eps:
            return (hi + lo) / 2
        mid = (hi + lo) / 2
        if fn(mid) < target:
            lo = mid
        else:
            hi = mid
    raise RuntimeError("binary_search did not converge in 500 halvings")


def get_coeffs(kernel_fn, order):
    """Taps k(s*[-order..order]) / k(0) with the spacing s that balances spatial
    and spectral coverage of the profile `kernel_fn` (a function of distance).
    Follows py:14-28 step by step (fp32 profile samples, numpy FFT)."""
    n = _GRID_POINTS
    x = np.linspace(-_GRID_HALF_WIDTH, _GRID_HALF_WIDTH, n)
    fn_values = kernel_fn(torch.from_numpy(x).float()).cpu().data.numpy()
    w = 2 * np.pi * np.fft.fftfreq(n, 2 * _GRID_HALF_WIDTH / n)
    fft_values = np.absolute(np.fft.fft(fn_values) / (2 * np.pi * np.sqrt(n)))

    def objective(spacing):
        return coverage_diff(spacing, order=order, x=x, w=w, fn_values=fn_values, fft_values=fft_values)

    s = binary_search(0, _SPACING_BOUNDS, objective, _SPACING_TOL)
    vals = kernel_fn(s * torch.arange(-order, order + 1).float())
    return vals / vals[order]


class DiscretizedKernelFN(nn.Module):
    """Forward taps of a profile k(d^2) and taps of dk/d(d^2) (py:162-181)."""

    def __init__(self, kernel_fn, order):
        super().__init__()
        self.kernel_fn = kernel_fn
        self.order = order
        self._forward_coeffs = get_coeffs(lambda dist: self.kernel_fn(dist ** 2), order).detach()

        def profile_derivative(dist):
            with torch.autograd.enable_grad():
                z = dist ** 2 + torch.zeros_like(dist, requires_grad=True)
                (g,) = torch.autograd.grad(self.kernel_fn(z).sum(), z)
            return g

        self._deriv_coeffs = get_coeffs(profile_derivative, order).detach()
        _log.info("discretized kernel taps %s, derivative taps %s",
                  self._forward_coeffs.tolist(), self._deriv_coeffs.tolist())

    def get_coeffs(self):
        return self._forward_coeffs

    def get_deriv_coeffs(self):
        return self._deriv_coeffs


# ----------------------------------------------------------------- profiles
#
# All profiles are functions of the SQUARED distance d2 (py:202-245).  Matern-nu, with t = sqrt(2 nu) * sqrt(d2):
#     nu = 1/2:  e^-t          nu = 3/2:  (1 + t) e^-t          nu = 5/2:  (1 + t + t^2 / 3) e^-t
# and d/d(d2) in closed form (finite at d2 = 0, which plain autograd through sqrt is not):
#     nu = 3/2:  -(3/2) e^-t                 nu = 5/2:  -(5/6) (1 + t) e^-t

def rbf(d2):
    """exp(-d^2) (py:202-203)."""
    return (-d2).exp()


def _matern_terms(d2, nu):
    """(distance, decay e^-t, polynomial factor) of the Matern-nu profile; written like the reference so that the
    fp32 results agree to the last bit (py:209-217, py:235-244)."""
    dist = d2.abs().sqrt()
    decay = torch.exp(-math.sqrt(2 * nu) * dist)
    if nu == 0.5:
        poly = 1
    elif nu == 1.5:
        poly = (math.sqrt(3) * dist).add(1)
    elif nu == 2.5:
        poly = (math.sqrt(5) * dist).add(1).add(5.0 / 3.0 * dist ** 2)
    else:
        raise NotImplementedError(f"Matern nu={nu}")
    return dist, decay, poly


class Matern(torch.autograd.Function):
    """Matern-nu profile in d^2, nu in {1.5, 2.5}, with the closed-form derivative above (py:207-232)."""

    _SLOPE = {1.5: lambda dist: -(3 / 2), 2.5: lambda dist: -(5 / 6) * (1 + dist * math.sqrt(5))}

    @staticmethod
    def forward(ctx, d2, nu):
        if nu not in Matern._SLOPE:
            raise NotImplementedError(f"Matern nu={nu}")
        dist, decay, poly = _matern_terms(d2, nu)
        if any(ctx.needs_input_grad):
            ctx.nu = nu
            ctx.save_for_backward(dist, decay)
        return poly * decay

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient with respect to nu")
        dist, decay = ctx.saved_tensors
        return grad_output * Matern._SLOPE[ctx.nu](dist) * decay, None


def matern(d2, nu=.5):
    """Plain-autograd Matern profile, nu in {0.5, 1.5, 2.5} (py:234-245)."""
    _, decay, poly = _matern_terms(d2, nu)
    return poly * decay
