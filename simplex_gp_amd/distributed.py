"""Point-sharded K.v over the GPUs of one node (new work, the
reference has no multi-GPU code).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The
N points are split into contiguous row blocks, one per rank.

    build   every rank builds the lattice from ALL positions (same vertex
            numbering on every rank: ids are the deterministic first-touch
            order of the shard-major lattice order), but its splat CSR / slice
            tables cover only its own rows
    splat   own rows -> full-size vertex accumulator values[m, vd]
    exchange ONE all-reduce (sum) of values over the group: the only collective
            on the data path; message = m * vd * 4 bytes
    blur    replicated on every rank
    slice   own rows

Output rows stay sharded; reductions over points (CG dot products) use
`all_reduce_sum` below.  The lattice object is duck-typed (build / new_values /
splat / blur / slice / m), so the choreography is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist

from .lattice import Lattice


def shard_bounds(n, world_size, rank):
    """Contiguous near-equal row blocks: the first n % world ranks get one extra row."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_sum(t, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_rows(t_local, group=None, extra=None):
    """Concatenate every rank's rows (row counts may differ) -> (tensor, counts).
    `extra`: a few integers per rank exchanged in the same small collective as the row counts
    (returned as a third value, one list per rank); one host synchronisation in total."""
    world = dist.get_world_size(group)
    meta = torch.tensor([t_local.shape[0]] + [int(e) for e in (extra or [])], dtype=torch.int64, device=t_local.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    table = torch.stack(metas, 0).tolist()
    counts = [int(row[0]) for row in table]
    biggest = max(counts)
    padded = torch.zeros((biggest,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device=t_local.device)
    padded[: t_local.shape[0]] = t_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = torch.cat([p[:k] for p, k in zip(parts, counts)], 0)
    if extra is None:
        return out, counts
    return out, counts, [[int(v) for v in row[1:]] for row in table]


class ShardedLatticeMVM:
    """K(ref_all) @ v with v and the result sharded by rows over the group.

    Two ways to build it:
      ShardedLatticeMVM(ref_all, coeffs)                 every rank holds all positions and builds the
                                                         whole vertex set itself (no communication)
      ShardedLatticeMVM.from_local_rows(ref_local, ...)  every rank holds only its rows; the per-rank vertex
                                                         keys are all-gathered once and merged (no replicated
                                                         O(N_total) work; same vertex numbering)
    """

    def __init__(self, ref_all, coeffs, group=None, lattice=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = ref_all.shape[0]
        self.lo, self.hi = shard_bounds(self.n, self.world, self.rank)
        self.lattice = lattice if lattice is not None else Lattice(ref_all.device)
        self.lattice.build(ref_all, coeffs, shard=(self.rank, self.world))
        self._vd = None
        self._values = self._scratch = None

    @classmethod
    def from_local_rows(cls, ref_local, coeffs, group=None, lattice=None, n_total=None):
        """Build from this rank's rows only (rank r must hold block r of shard_bounds(n_total, world, r))."""
        self = cls.__new__(cls)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lattice = lattice if lattice is not None else Lattice(ref_local.device)
        self._vd = None
        self._values = self._scratch = None
        if self.world == 1:
            self.n = ref_local.shape[0]
            self.lo, self.hi = 0, self.n
            self.lattice.build(ref_local, coeffs)
            return self
        keys = self.lattice.build_local(ref_local, coeffs)
        # the one exchange of the build: vertex keys, with the key and row counts riding in its size message
        all_keys, counts, rows = all_gather_rows(keys, group, extra=[ref_local.shape[0]])
        self.lattice.build_merge(all_keys, counts, self.rank)
        every = [r[0] for r in rows]
        self.n = sum(every) if n_total is None else n_total
        self.lo = sum(every[: self.rank])
        self.hi = self.lo + every[self.rank]
        return self

    @property
    def m(self):
        return self.lattice.m

    def local_rows(self, t):
        """The rows of a full-size tensor that this rank owns."""
        return t[self.lo:self.hi]

    def _workspace(self, vd):
        if self._vd != vd:
            self._values = self.lattice.new_values(vd)
            self._scratch = self.lattice.new_values(vd)
            self._vd = vd
        return self._values, self._scratch

    def matmul(self, v_local, out=None):
        if v_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} owns rows [{self.lo}, {self.hi}) but got {v_local.shape[0]} rows")
        squeeze = v_local.dim() == 1
        if squeeze:
            v_local = v_local.unsqueeze(-1)
        vd = v_local.shape[1]
        values, scratch = self._workspace(vd)
        self.lattice.splat(v_local, values)
        all_reduce_sum(values, self.group)               # the one exchange step
        blurred = self.lattice.blur(values, scratch, vd=vd)
        res = self.lattice.slice(blurred, out, vd=vd)
        return res.squeeze(-1) if squeeze else res

    __call__ = matmul

    def gather_rows(self, t_local):
        """All ranks' row blocks concatenated (for tests / small outputs)."""
        if self.world == 1:
            return t_local
        return all_gather_rows(t_local.contiguous(), self.group)[0]
