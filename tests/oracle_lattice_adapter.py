"""TEST HELPER: an oracle-backed object with the duck type of simplex_gp_amd.Lattice
(build / new_values / splat / blur / slice / m), CPU tensors.  Lets the sharded
choreography in simplex_gp_amd.distributed run under gloo without a GPU."""
import numpy as np
import torch

from oracle import oracle


class OracleLattice:
    def __init__(self, device=None):
        self._o = None

    def build(self, ref, coeffs, shard=None):
        oracle.set_exact_mode(False)      # the duplicate-free lattice, like the HIP path
        try:
            self._o = oracle.Lattice(ref.numpy(), np.asarray(coeffs, np.float32))
        finally:
            oracle.set_exact_mode(True)
        self.n = ref.shape[0]
        from simplex_gp_amd.distributed import shard_bounds
        self.lo, self.hi = (0, self.n) if shard is None else shard_bounds(self.n, shard[1], shard[0])
        return self

    @property
    def m(self):
        return self._o.m

    def new_values(self, vd):
        return torch.empty((self.m, vd), dtype=torch.float32)

    def splat(self, src, values=None):
        full = np.zeros((self.n, src.shape[1]), np.float32)
        full[self.lo:self.hi] = src.numpy()
        res = torch.from_numpy(self._o.splat(full))
        if values is None:
            return res
        values.copy_(res)
        return values

    def blur(self, values, scratch=None, vd=None):
        res = torch.from_numpy(self._o.blur(values.numpy()))
        target = scratch if scratch is not None else values     # d+1 odd/even does not matter here
        target.copy_(res)
        return target

    def slice(self, values, out=None, vd=None):
        res = torch.from_numpy(self._o.slice(values.numpy())[self.lo:self.hi].copy())
        if out is None:
            return res
        out.copy_(res)
        return out
