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


# A group of one rank needs no communication and normally issues none.  FORCE_COLLECTIVES = True makes every collective
# below run even then (tests/checks/rccl_world1.py: one rank on one GPU still drives RCCL through the exact calls an
# 8-GPU job makes -- init with device_id, all_gather of keys, all_reduce of the vertex values, barrier with device_ids).
FORCE_COLLECTIVES = False


def _collective_needed(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def all_reduce_sum(t, group=None):
    if _collective_needed(group):
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
        self._n_total = n_total
        self.rebuild(ref_local, coeffs)
        return self

    def rebuild(self, ref_local, coeffs):
        """Re-run the sharded build for new positions / taps on the same lattice object (device buffers are
        recycled): local structure, ONE all-gather of the per-rank vertex keys, merge.  Used by every
        hyper-parameter step of a training loop and by bench.py's rebuild cadence."""
        self._vd = None                      # m may change: the accumulators are re-allocated on the next MVM
        if self.world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
            self.n = ref_local.shape[0]
            self.lo, self.hi = 0, self.n
            self.lattice.build(ref_local, coeffs)
            return self
        keys = self.lattice.build_local(ref_local, coeffs)
        # the one exchange of the build: vertex keys, with the key and row counts riding in its size message
        all_keys, counts, rows = all_gather_rows(keys, self.group, extra=[ref_local.shape[0]])
        every = [r[0] for r in rows]
        self.lattice.build_merge(all_keys, counts, self.rank, total_points=sum(every))
        self.n = sum(every) if self._n_total is None else self._n_total
        self.lo = sum(every[: self.rank])
        self.hi = self.lo + every[self.rank]
        self.key_bytes_exchanged = int(all_keys.numel() * all_keys.element_size())
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

    def matmul(self, v_local, out=None, marks=None):
        """One sharded MVM.  `marks`: an optional list that receives a callable-produced stamp after each of the
        four stages (splat, exchange, blur, slice) -- bench.py passes a recorder of device events."""
        if v_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} owns rows [{self.lo}, {self.hi}) but got {v_local.shape[0]} rows")
        squeeze = v_local.dim() == 1
        if squeeze:
            v_local = v_local.unsqueeze(-1)
        vd = v_local.shape[1]
        values, scratch = self._workspace(vd)
        mark = marks if marks is not None else (lambda name: None)
        mark("start")
        self.lattice.splat(v_local, values)
        mark("splat")
        all_reduce_sum(values, self.group)               # the one exchange step
        mark("exchange")
        blurred = self.lattice.blur(values, scratch, vd=vd)
        mark("blur")
        res = self.lattice.slice(blurred, out, vd=vd)
        mark("slice")
        return res.squeeze(-1) if squeeze else res

    def exchange_bytes(self, vd):
        """Payload of the per-MVM all-reduce: the vertex accumulator, m x values_stride(vd) floats."""
        return int(self.m) * int(self.lattice.values_stride(vd)) * 4

    __call__ = matmul

    def gather_rows(self, t_local):
        """All ranks' row blocks concatenated (for tests / small outputs)."""
        if self.world == 1:
            return t_local
        return all_gather_rows(t_local.contiguous(), self.group)[0]


def sharded_solve(op, rhs_local, outputscale, noise, **cg_args):
    """(s K + sigma^2 I)^-1 rhs over a row-sharded operator `op` (a ShardedLatticeMVM): batched CG whose MVM is the
    sharded one (own-row splat, vertex all-reduce, blur, own-row slice) and whose dot products are summed over the
    ranks.  rhs_local / the result are this rank's rows.  Every rank must call it (collectives inside)."""
    from .solvers import batched_cg
    s, noise = float(outputscale), float(noise)

    def mm(V):
        return op.matmul(V).mul_(s).add_(V, alpha=noise)
    return batched_cg(mm, rhs_local, reduce=lambda t: all_reduce_sum(t, op.group), **cg_args)
