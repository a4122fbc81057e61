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

def rbf(d2):
    """exp(-d^2) (py:202-203)."""
    return torch.exp(-d2)


class Matern(torch.autograd.Function):
    """Matern-nu profile in d^2 with a closed-form derivative that stays finite
    at d = 0 (py:207-232).  nu in {1.5, 2.5}."""

    @staticmethod
    def forward(ctx, d2, nu):
        dist = d2.abs().sqrt()
        decay = torch.exp(-math.sqrt(nu * 2) * dist)
        if nu == 1.5:
            poly = (math.sqrt(3) * dist).add(1)
        elif nu == 2.5:
            poly = (math.sqrt(5) * dist).add(1).add(5.0 / 3.0 * dist ** 2)
        else:
            raise NotImplementedError(f"Matern nu={nu}")
        if any(ctx.needs_input_grad):
            ctx.nu = nu
            ctx.save_for_backward(dist, decay)
        return poly * decay

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient with respect to nu")
        dist, decay = ctx.saved_tensors
        if ctx.nu == 1.5:
            factor = -(3 / 2)
        elif ctx.nu == 2.5:
            factor = -(5 / 6) * (1 + dist * math.sqrt(5))
        else:
            raise NotImplementedError
        return grad_output * factor * decay, None


def matern(d2, nu=.5):
    """Plain-autograd Matern profile, nu in {0.5, 1.5, 2.5} (py:234-245)."""
    dist = d2.abs().sqrt()
    decay = torch.exp(-math.sqrt(nu * 2) * dist)
    if nu == 0.5:
        poly = 1
    elif nu == 1.5:
        poly = (math.sqrt(3) * dist).add(1)
    elif nu == 2.5:
        poly = (math.sqrt(5) * dist).add(1).add(5.0 / 3.0 * dist ** 2)
    else:
        raise NotImplementedError(f"Matern nu={nu}")
    return poly * decay
