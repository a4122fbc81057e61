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


LAST_GATHER = {}          # the most recent all_gather_rows on this rank: {"us", "bytes_out", "world"}; "us" only under TIME_GATHER
TIME_GATHER = False       # bench.py: synchronise the device around all_gather_rows and time it (a measuring mode)


def _gather_into_tensor_supported(group=None):
    """Whether all_gather_rows uses all_gather_into_tensor.  A property of the torch build and of the group's back end
    (gloo and nccl = RCCL both have it in this torch), the same on every rank of a group: the choice must never depend
    on whether a collective raised on THIS rank (a one-rank error would pair its list-form all_gather with the peers'
    tensor form: a mismatch or a deadlock in place of the error)."""
    if not hasattr(dist, "all_gather_into_tensor"):
        return False
    return str(dist.get_backend(group)).lower() in ("nccl", "gloo")


def all_gather_rows(t_local, group=None, extra=None):
    """Concatenate every rank's rows (row counts may differ) -> (tensor, counts).
    `extra`: a few integers per rank exchanged in the same small collective as the row counts
    (returned as a third value, one list per rank).

    Two collectives: the counts (they have to reach the host anyway: plx_build_merge takes them as host integers, and
    they size the buffers), then ONE all_gather_into_tensor of the rows, padded to the largest count, into one
    [world, biggest, ...] buffer.  (Rounds 2-4 used the list forms: `world` output tensors per collective and a zero-filled
    padded copy; the list form stays as the fallback for back ends without all_gather_into_tensor.)"""
    import time
    if TIME_GATHER and t_local.is_cuda:
        torch.cuda.synchronize(t_local.device)
    t0 = time.perf_counter()
    world = dist.get_world_size(group)
    meta = torch.tensor([t_local.shape[0]] + [int(e) for e in (extra or [])], dtype=torch.int64, device=t_local.device)
    # (outputs in the CONCATENATED form [world * k, ...]: the one shape both gloo and RCCL accept; viewed as [world, k, ...])
    metas = torch.empty((world * meta.shape[0],), dtype=meta.dtype, device=meta.device)
    into_tensor = _gather_into_tensor_supported(group)       # decided once, identically on every rank: never per call
    if into_tensor:
        dist.all_gather_into_tensor(metas, meta, group=group)
        table = metas.view(world, -1).tolist()
    else:
        lst = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(lst, meta, group=group)
        table = torch.stack(lst, 0).tolist()
    counts = [int(row[0]) for row in table]
    biggest = max(counts)
    if all(c == biggest for c in counts):
        padded = t_local.contiguous()                       # equal counts: nothing to pad, nothing to copy
    else:
        padded = torch.empty((biggest,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device=t_local.device)
        padded[: t_local.shape[0]] = t_local
        padded[t_local.shape[0]:] = 0
    gathered = torch.empty((world * biggest,) + tuple(padded.shape[1:]), dtype=padded.dtype, device=padded.device)
    if into_tensor:
        dist.all_gather_into_tensor(gathered, padded, group=group)
        gathered = gathered.view((world,) + tuple(padded.shape))
    else:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        gathered = torch.stack(parts, 0)
    if all(c == biggest for c in counts):
        out = gathered.reshape((world * biggest,) + tuple(padded.shape[1:]))
    else:
        out = torch.cat([gathered[r, :k] for r, k in enumerate(counts)], 0)
    if TIME_GATHER and out.is_cuda:
        torch.cuda.synchronize(out.device)
    LAST_GATHER.update(us=(time.perf_counter() - t0) * 1e6 if TIME_GATHER else None,
                       bytes_out=out.numel() * out.element_size(), world=world)
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


# ----------------------------------------------------------------------------------------------------------------------
# Column-sharded batched CG, and the points x columns grid.
#
# The columns of a batched solve -- [y | probe vectors] in the reference's training loop
# (experiments/train_simplexgp.py:34-41: one mBCG call over 1 + num_trace_samples columns) -- never interact: batched CG
# freezes every column on its own residual.  So the cheapest way to put several GPUs on ONE solve is to give each rank
# the whole lattice (replicated build: every rank holds all positions) and a share of the COLUMNS.  There is then no
# collective inside the iteration at all -- no vertex all-reduce, no dot-product all-reduce -- and one all-gather of the
# solution (and the Lanczos coefficients) at the end.  Point sharding (ShardedLatticeMVM above) pays one all-reduce of
# values[m, vd] per MVM and replicates the blur; it is the mode for operators too large for one GPU's time budget at
# vd = 1.  The grid composes the two: C column groups x P point shards, the vertex all-reduce confined to the P ranks of
# a column group.


def column_bounds(t, parts, index):
    """Contiguous near-equal column blocks (the first t % parts blocks get one extra column)."""
    return shard_bounds(t, parts, index)


def all_gather_columns(x_local, t, group=None):
    """Every rank holds columns column_bounds(t, world, rank) of a [n, t] matrix (the same rows everywhere): returns the
    whole matrix on every rank.  One collective; blocks are padded to the widest."""
    if not _collective_needed(group):
        return x_local
    world = dist.get_world_size(group)
    if world == 1 and not FORCE_COLLECTIVES:
        return x_local
    widest = -(-t // world)
    n = x_local.shape[0]
    pad = x_local.new_zeros(n, widest)
    pad[:, : x_local.shape[1]] = x_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous(), group=group)
    cols = []
    for r, p in enumerate(parts):
        lo, hi = column_bounds(t, world, r)
        cols.append(p[:, : hi - lo])
    return torch.cat(cols, 1)


def column_sharded_solve(solve, rhs, group=None, gather=True):
    """Batched solve with the right-hand-side columns split over the ranks of `group`.

    solve(rhs_block) -> (x_block, info): the single-process solver over the WHOLE operator (every rank has built the
    same lattice), e.g. `lambda B: model.khat_solve(x, B, K=K, max_iter=..., tol=...)`.  rhs: the full [n, t] matrix,
    identical on every rank (only this rank's block of columns is read).  Returns (X, info): with gather=True the whole
    [n, t] solution on every rank, info["residual"] / ["tridiag"] / ["rz0"] for all t columns and
    info["iterations"] = the most any rank needed; with gather=False this rank's block and its own info (the caller
    keeps working column-sharded, e.g. a training step that all-reduces hyper-parameter gradients only).
    A rank with no column (world > t) solves nothing and contributes nothing."""
    world = dist.get_world_size(group) if _collective_needed(group) else 1
    rank = dist.get_rank(group) if _collective_needed(group) else 0
    t = rhs.shape[1]
    lo, hi = column_bounds(t, world, rank)
    info = {"iterations": 0}
    if hi > lo:
        x_local, info = solve(rhs[:, lo:hi].contiguous())
    else:
        x_local = rhs.new_zeros(rhs.shape[0], 0)
    info = dict(info, columns=(lo, hi), exchange_bytes=0)
    if not gather or world == 1:
        return x_local, info
    X = all_gather_columns(x_local, t, group)
    out = {"columns": (lo, hi), "exchange_bytes": int(rhs.shape[0]) * int(t) * rhs.element_size()}
    it = torch.tensor([int(info["iterations"])], dtype=torch.int64, device=rhs.device)
    dist.all_reduce(it, op=dist.ReduceOp.MAX, group=group)
    out["iterations"] = int(it.item())
    for key in ("residual", "rz0"):
        if _any_has(info, key, group, rhs.device):
            v = info.get(key)
            v = v.reshape(1, -1).to(rhs.dtype) if v is not None else rhs.new_zeros(1, 0)
            out[key] = all_gather_columns(v, t, group).reshape(-1)
    if _any_has(info, "tridiag", group, rhs.device):
        # [t_local, k, k] with k = this rank's iteration count: pad to the common k (identity rows, as
        # solvers._tridiag_from_cg freezes converged columns), gather along the column axis
        k = out["iterations"]
        T = info.get("tridiag")
        if T is None:
            T = torch.zeros(0, k, k, dtype=torch.float64, device=rhs.device)
        if T.shape[-1] < k:
            big = torch.eye(k, dtype=T.dtype, device=T.device).repeat(T.shape[0], 1, 1)
            big[:, : T.shape[-1], : T.shape[-1]] = T
            T = big
        flat = all_gather_columns(T.reshape(T.shape[0], k * k).t().contiguous(), t, group)     # [k*k, t]
        out["tridiag"] = flat.t().reshape(t, k, k)
    return X, out


def _any_has(info, key, group, device):
    """True when any rank's info carries `key` (a rank without columns has an empty info)."""
    flag = torch.tensor([1 if info.get(key) is not None else 0], dtype=torch.int64, device=device)
    if _collective_needed(group):
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(flag.item())


class SolveGrid:
    """C column groups x P point shards over world = C * P ranks: rank = c * P + p.

    point_group   the P ranks that share a column block (same c): they run ONE row-sharded operator together
                  (ShardedLatticeMVM over this group: vertex all-reduce and CG dot products stay inside it)
    column_group  the C ranks that hold the same rows (same p): the final all-gather of columns runs over it
    P = 1 is pure column sharding, C = 1 pure point sharding."""

    def __init__(self, column_groups, world=None, rank=None):
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        if column_groups < 1 or self.world % column_groups:
            raise ValueError(f"{column_groups} column groups do not divide {self.world} ranks")
        self.C, self.P = column_groups, self.world // column_groups
        self.c, self.p = self.rank // self.P, self.rank % self.P
        self.point_group = self.column_group = None
        if dist.is_available() and dist.is_initialized() and self.world > 1:
            # every rank creates every group, in the same order (torch.distributed's rule)
            for c in range(self.C):
                g = dist.new_group([c * self.P + p for p in range(self.P)])
                if c == self.c:
                    self.point_group = g
            for p in range(self.P):
                g = dist.new_group([c * self.P + p for c in range(self.C)])
                if p == self.p:
                    self.column_group = g

    def columns(self, t):
        return column_bounds(t, self.C, self.c)

    def rows(self, n):
        return shard_bounds(n, self.P, self.p)

    def solve(self, op, rhs_rows, outputscale, noise, gather=True, **cg_args):
        """(s K + sigma^2 I)^-1 rhs on the grid.  op: a ShardedLatticeMVM over self.point_group; rhs_rows: this rank's
        rows of ALL t columns.  Returns this rank's rows of the solution -- all t columns (gather=True: one all-gather over
        the column group) or its own column block."""
        t = rhs_rows.shape[1]
        lo, hi = self.columns(t)
        if hi > lo:
            x_local, info = sharded_solve(op, rhs_rows[:, lo:hi].contiguous(), outputscale, noise, **cg_args)
        else:
            x_local, info = rhs_rows.new_zeros(rhs_rows.shape[0], 0), {"iterations": 0}
        if not gather or self.C == 1:
            return x_local, info
        return all_gather_columns(x_local, t, self.column_group), info


# ----------------------------------------------------------------------------------------------------------------------
# A column-sharded TRAINING STEP: the marginal likelihood of solvers.marginal_log_likelihood with its [y | probes] columns
# split over the ranks.  Every rank holds all positions, builds the same lattice (and, with pre_size > 0, the same
# preconditioner) and draws the same probe vectors from the same seed; it solves its own columns, forms its own share of
# the value (the quadratic term lives with the rank that has column 0, the stochastic log-determinant is a mean over the
# probe columns) and of the surrogate whose gradient is the gradient of the likelihood, and runs its own backward pass --
# ONE all-reduce of two scalars for the value, one all-reduce of the hyper-parameter gradients (d + 3 numbers) per step, no
# collective inside the solve or the backward filter.


def column_sharded_mll(model, x, y, num_probes=10, max_cg_iter=1000, cg_tol=1e-4, seed=0, pre_size=0, group=None):
    """Per-datapoint log marginal likelihood (as solvers.marginal_log_likelihood) with the columns of its batched solve
    sharded over `group`.  The returned scalar has the same VALUE on every rank; its backward() leaves THIS RANK'S SHARE
    of the gradient in the parameters' .grad -- call all_reduce_gradients(model, group) before the optimiser step."""
    import math
    from . import solvers
    world = dist.get_world_size(group) if _collective_needed(group) else 1
    rank = dist.get_rank(group) if _collective_needed(group) else 0
    n = y.shape[0]
    r = (y - model.mean).reshape(-1, 1)
    gen = torch.Generator(device=y.device).manual_seed(seed)          # the same probes on every rank
    K = model.kernel(x, x)
    mm = model.khat_matmul(x, K)
    precond = model.preconditioner(x, pre_size, K=K) if pre_size > 0 else None
    t = 1 + num_probes
    lo, hi = column_bounds(t, world, rank)
    # [r^T u, sum over local probes of weight_i * quadrature_i, logdet(P) as rank 0 computed it]
    stats = torch.zeros(3, dtype=torch.float64, device=y.device)
    surrogate = None
    info = {"iterations": 0}
    with torch.no_grad():
        if precond is None:
            Z = torch.randint(0, 2, (n, num_probes), generator=gen, device=y.device).to(r.dtype) * 2 - 1
        else:
            Z = precond.sample(num_probes, generator=gen)
        rhs = torch.cat([r.detach(), Z], 1)
    if hi > lo:
        with torch.no_grad():
            sol, info = model.khat_solve(x, rhs[:, lo:hi].contiguous(), K=K, max_iter=max_cg_iter, tol=cg_tol, want_tridiag=True,
                                         precond=precond)
            has_y = lo == 0
            p0 = 1 if has_y else 0                                    # first probe column inside the local block
            W = sol[:, p0:]
            terms = solvers.slq_terms(info.get("tridiag_host", info["tridiag"])[p0:]).to(stats.device)
            if terms.numel():
                weights = torch.full_like(terms, float(n)) if precond is None else info["rz0"][p0:].double()
                stats[1] = (weights * terms).sum()
            Zl = Z[:, lo + p0 - 1: hi - 1]                            # the probes this rank solved for
            if precond is not None and Zl.shape[1]:
                Zl = precond.solve(Zl)
            if has_y:
                u = sol[:, :1]
                stats[0] = (r.detach() * u).sum().double()
        cols = ([u] if has_y else []) + ([Zl] if Zl.shape[1] else [])
        KV = mm(torch.cat(cols, 1))                                   # differentiable MVM over the local columns
        surrogate = torch.zeros((), dtype=KV.dtype, device=KV.device)
        if has_y:
            surrogate = surrogate - (u * r).sum() + 0.5 * (u * KV[:, :1]).sum()
        if Zl.shape[1]:
            surrogate = surrogate - 0.5 * (W * KV[:, p0:]).sum() / num_probes
    # logdet(P) travels inside the reduced statistics, contributed by ONE rank, so that every rank adds the same number:
    # the replicated factors are built by the same deterministic kernels from the same inputs, but nothing else in the
    # estimator relies on that, and a rank whose pivots differed would otherwise return a different value silently
    if precond is not None and rank == 0:
        stats[2] = float(precond.logdet())
    all_reduce_sum(stats, group)
    logdet = stats[1] / num_probes + stats[2]
    value = -0.5 * stats[0] - 0.5 * logdet - 0.5 * n * math.log(2 * math.pi)
    if surrogate is None:                                             # a rank without columns: value only, no gradient share
        surrogate = (model.mean * 0.0).sum()
    out = (value.to(surrogate.dtype) + (surrogate - surrogate.detach())) / n
    out.cg_info = info
    return out


def all_reduce_gradients(model, group=None):
    """Sum the hyper-parameter gradients over the ranks (after column_sharded_mll(...).backward()): one collective over a
    flat buffer of all gradients."""
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    if not _collective_needed(group):
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        k = p.grad.numel()
        p.grad.copy_(flat[off:off + k].reshape(p.grad.shape))
        off += k
