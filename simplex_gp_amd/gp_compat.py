"""The two GPyTorch base classes the operator surface needs.

The reference subclasses gpytorch.kernels.Kernel and gpytorch.lazy.LazyTensor
(bilateral_kernel.py:9-10).  When a GPyTorch that still ships
`gpytorch.lazy.LazyTensor` (<= 1.8) is importable, those are used and the
lattice kernels plug straight into ExactGP / mBCG.  GPyTorch is not installed
in the build image, so otherwise small stand-alone bases with the same
protocol (the methods the reference overrides: _matmul, _size,
_transpose_nonbatch, diag; Kernel.forward with a softplus lengthscale) keep the
operator usable with simplex_gp_amd.solvers.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

try:  # pragma: no cover - exercised only where gpytorch is installed
    from gpytorch.kernels import Kernel as _GKernel
    from gpytorch.lazy import LazyTensor as _GLazyTensor
    HAVE_GPYTORCH = True
except Exception:  # noqa: BLE001
    _GKernel = _GLazyTensor = None
    HAVE_GPYTORCH = False


def _inv_softplus(x):
    return x + torch.log(-torch.expm1(-x))


class _Kernel(nn.Module):
    """Minimal stand-in for gpytorch.kernels.Kernel: ARD lengthscale with a
    softplus (Positive) constraint, raw parameter initialised to 0 so that the
    initial lengthscale is softplus(0) = 0.6931 like GPyTorch's default."""

    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None,
                 lengthscale_prior=None, lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self.ard_num_dims = ard_num_dims
        self.batch_shape = batch_shape
        self.active_dims = active_dims
        self.eps = eps
        if self.has_lengthscale:
            num = 1 if ard_num_dims is None else ard_num_dims
            self.raw_lengthscale = nn.Parameter(torch.zeros(*batch_shape, 1, num))

    @property
    def lengthscale(self):
        return F.softplus(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        value = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype, device=self.raw_lengthscale.device)
        with torch.no_grad():
            self.raw_lengthscale.copy_(_inv_softplus(value.expand_as(self.raw_lengthscale)))

    def forward(self, x1, x2, diag=False, **params):
        raise NotImplementedError

    def __call__(self, x1, x2=None, diag=False, **params):
        if x2 is None:
            x2 = x1
        if x1.dim() == 1:
            x1 = x1.unsqueeze(-1)
        if x2.dim() == 1:
            x2 = x2.unsqueeze(-1)
        if self.active_dims is not None:
            x1 = x1.index_select(-1, torch.as_tensor(self.active_dims, device=x1.device))
            x2 = x2.index_select(-1, torch.as_tensor(self.active_dims, device=x2.device))
        return self.forward(x1, x2, diag=diag, **params)


class _LazyTensor:
    """Minimal stand-in for gpytorch.lazy.LazyTensor: a matrix known only
    through matmul."""

    def __init__(self, *args, **kwargs):
        self._args = args
        self._kwargs = kwargs

    # -- protocol the subclasses implement
    def _matmul(self, rhs):
        raise NotImplementedError

    def _size(self):
        raise NotImplementedError

    def _transpose_nonbatch(self):
        raise NotImplementedError

    # -- derived surface
    def size(self, dim=None):
        s = self._size()
        return s if dim is None else s[dim]

    @property
    def shape(self):
        return self._size()

    def dim(self):
        return len(self._size())

    @property
    def dtype(self):
        return self._args[0].dtype

    @property
    def device(self):
        return self._args[0].device

    def matmul(self, rhs):
        if rhs.dim() == 1:
            return self._matmul(rhs.unsqueeze(-1)).squeeze(-1)
        return self._matmul(rhs)

    def __matmul__(self, rhs):
        return self.matmul(rhs)

    def transpose(self, a=-1, b=-2):
        return self._transpose_nonbatch()

    def t(self):
        return self._transpose_nonbatch()

    def diag(self):
        raise NotImplementedError

    def evaluate(self):
        n = self.size(-1)
        return self.matmul(torch.eye(n, dtype=self.dtype, device=self.device))


Kernel = _GKernel if HAVE_GPYTORCH else _Kernel
LazyTensor = _GLazyTensor if HAVE_GPYTORCH else _LazyTensor
