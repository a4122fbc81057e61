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

    # -- sharded build (plx_build_local / plx_build_merge): local structure -> vertex keys -> union numbering.
    # Vertex ids follow the rank-major first occurrence of the keys, like libplx's merge.
    @staticmethod
    def _pack(keys16):
        """[m, d] int16 -> [m, ceil(d/2)] int32, two coordinates per word (the layout plx_copy_local_keys emits)."""
        m, d = keys16.shape
        dw = (d + 1) // 2
        padded = np.zeros((m, 2 * dw), np.uint16)
        padded[:, :d] = keys16.view(np.uint16)
        return (padded[:, 0::2].astype(np.uint32) | (padded[:, 1::2].astype(np.uint32) << 16)).view(np.int32)

    def build_local(self, ref_local, coeffs):
        oracle.set_exact_mode(False)
        try:
            self._local = oracle.Lattice(ref_local.numpy(), np.asarray(coeffs, np.float32))
        finally:
            oracle.set_exact_mode(True)
        self._ref_local, self._coeffs = ref_local, np.asarray(coeffs, np.float32)
        return torch.from_numpy(self._pack(self._local.keys).copy())

    def build_merge(self, all_keys, counts, rank, total_points=0):
        keys = all_keys.numpy()
        assert keys.shape[0] == sum(counts) and counts[rank] == self._local.m
        first = {}
        for row in keys:                                   # first occurrence in rank order numbers the union
            first.setdefault(row.tobytes(), len(first))
        off = sum(counts[:rank])
        self._to_global = np.array([first[row.tobytes()] for row in keys[off:off + counts[rank]]], np.int64)
        self._m_union = len(first)
        self._union_keys = np.frombuffer(b"".join(first.keys()), np.int32).reshape(len(first), keys.shape[1])
        self.n = self._ref_local.shape[0]
        self.lo, self.hi = 0, self.n
        self._merged = True
        return self

    @property
    def m(self):
        return self._m_union if getattr(self, "_merged", False) else self._o.m

    @staticmethod
    def values_stride(vd):
        return vd

    def new_values(self, vd):
        return torch.empty((self.m, vd), dtype=torch.float32)

    def splat(self, src, values=None):
        if getattr(self, "_merged", False):
            res = torch.zeros((self._m_union, src.shape[1]), dtype=torch.float32)
            res[torch.from_numpy(self._to_global)] = torch.from_numpy(self._local.splat(src.numpy()))
            if values is None:
                return res
            values.copy_(res)
            return values
        full = np.zeros((self.n, src.shape[1]), np.float32)
        full[self.lo:self.hi] = src.numpy()
        res = torch.from_numpy(self._o.splat(full))
        if values is None:
            return res
        values.copy_(res)
        return values

    def blur(self, values, scratch=None, vd=None):
        if getattr(self, "_merged", False):
            res = torch.from_numpy(self._blur_union(values.numpy()))
            target = scratch if scratch is not None else values
            target.copy_(res)
            return target
        res = torch.from_numpy(self._o.blur(values.numpy()))
        target = scratch if scratch is not None else values     # d+1 odd/even does not matter here
        target.copy_(res)
        return target

    def _blur_union(self, vals):
        """h:526-556 over the union vertex set, neighbours found through the packed keys."""
        keys = self._union_keys.view(np.uint16).reshape(self._m_union, -1).view(np.int16).astype(np.int64)
        d = self._ref_local.shape[1]
        keys = keys[:, :d]
        index = {k.tobytes(): i for i, k in enumerate(keys)}
        taps = self._coeffs
        r = len(taps) // 2
        cur = vals.astype(np.float32).copy()
        for axis in range(d + 1):
            new = np.zeros_like(cur)
            for t in range(-r, r + 1):
                nk = keys - t
                if axis < d:
                    nk[:, axis] = keys[:, axis] + t * d
                ids = np.array([index.get(k.tobytes(), -1) for k in nk], np.int64)
                ok = ids >= 0
                new[ok] += np.float32(taps[t + r]) * cur[ids[ok]]
            cur = new
        return cur

    def slice(self, values, out=None, vd=None):
        if getattr(self, "_merged", False):
            res = torch.from_numpy(self._local.slice(values.numpy()[self._to_global]))
            if out is None:
                return res
            out.copy_(res)
            return out
        res = torch.from_numpy(self._o.slice(values.numpy())[self.lo:self.hi].copy())
        if out is None:
            return res
        out.copy_(res)
        return out
