"""The caller of the hot path: batched CG + stochastic Lanczos log-det + GP
marginal likelihood, written against the operator surface only (`matmul`).

GPyTorch (ExactGP, ExactMarginalLogLikelihood, mBCG) is third-party and not
installed in the build image; this module is the minimal harness that drives
the lattice operator the way the reference's scripts do:
    tests/train_snelson.py:48-76            (config 1: Snelson, 100 Adam steps)
    experiments/train_simplexgp.py:29-57    (config 3/5: CG-based MLL training)
Every CG iteration is one K.v MVM with vd = 1 + num_probes on a lattice that is
built once per hyper-parameter setting (lattice_kernel.cached_filter).

batched_cg also runs over a row-sharded operator: pass the sharded `matmul` and
`reduce=distributed.all_reduce_sum` so that the dot products are summed over
ranks (distributed.sharded_solve).  The marginal likelihood below is
single-process.
"""
import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lattice_kernel import LatticeAccelerated


_dot_work = {}


def _colsum(a, b, reduce=None):
    """Column-wise <a, b> of two [n, t] matrices (summed over ranks through `reduce`).
    On the GPU this is libplx's plx_coldot (torch's (a*b).sum(0) is ~60x slower on
    tall, narrow row-major matrices); elsewhere plain torch."""
    if (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2
            and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.shape[1] <= 256):
        import ctypes
        from . import _native as nv
        n, t = a.shape
        key = (a.device.index, t)
        work = _dot_work.get(key)
        if work is None:
            work = _dot_work[key] = torch.empty(int(nv.lib().plx_coldot_work_floats(t)), dtype=torch.float32,
                                                device=a.device)
        out = torch.empty(t, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            rc = nv.lib().plx_coldot(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, t,
                                     ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(work.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
        nv.check(rc, "plx_coldot")
        s = out
    else:
        s = (a * b).sum(0)
    return reduce(s) if reduce is not None else s


def _native_ok(*ts):
    t0 = ts[0]
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous() and t.shape == t0.shape
               for t in ts) and t0.shape[1] <= 256


def _cg_update(X, R, P, AP, alpha, reduce=None):
    """X += P*alpha; R -= AP*alpha; returns the new column-wise |R|^2 (one fused pass on the GPU)."""
    if _native_ok(X, R, P, AP):
        import ctypes
        from . import _native as nv
        n, t = X.shape
        key = (X.device.index, t)
        work = _dot_work.get(key)
        if work is None:
            work = _dot_work[key] = torch.empty(int(nv.lib().plx_coldot_work_floats(t)), dtype=torch.float32,
                                                device=X.device)
        out = torch.empty(t, dtype=torch.float32, device=X.device)
        alpha = alpha.to(torch.float32).contiguous()
        p = lambda a: ctypes.c_void_p(a.data_ptr())          # noqa: E731
        with torch.cuda.device(X.device):
            rc = nv.lib().plx_cg_update(p(X), p(R), p(P), p(AP), p(alpha), n, t, p(out), p(work),
                                        ctypes.c_void_p(torch.cuda.current_stream(X.device).cuda_stream))
        nv.check(rc, "plx_cg_update")
        return reduce(out) if reduce is not None else out
    X.addcmul_(P, alpha)
    R.addcmul_(AP, -alpha)
    return _colsum(R, R, reduce)


def _cg_direction(P, R, beta):
    """P = R + P*beta in one pass."""
    if _native_ok(P, R):
        import ctypes
        from . import _native as nv
        beta = beta.to(torch.float32).contiguous()
        with torch.cuda.device(P.device):
            rc = nv.lib().plx_cg_direction(ctypes.c_void_p(P.data_ptr()), ctypes.c_void_p(R.data_ptr()),
                                           ctypes.c_void_p(beta.data_ptr()), P.shape[0], P.shape[1],
                                           ctypes.c_void_p(torch.cuda.current_stream(P.device).cuda_stream))
        nv.check(rc, "plx_cg_direction")
        return
    P.mul_(beta).add_(R)


class PivotedCholeskyPreconditioner:
    """P = L L^T + sigma^2 I with L [n, k] the rank-k pivoted Cholesky factor of s K: the
    preconditioner GPyTorch builds for (s K + sigma^2 I) when max_preconditioner_size > 0
    (experiments/train_simplexgp.py:34-41 runs with pre_size = 100, configs/simplexgp.yml).

    The factor needs the diagonal of K (ones: py:139-140) and k rows of it; a row is one
    MVM with a one-hot right-hand side (the operator is known through matmul only).  The
    pivot index stays on the device (scatter with an index tensor), so the k steps queue
    without host synchronisation.

    solve(R)  = P^-1 R by Woodbury, (R - L C^-1 L^T R) / sigma^2 with C = sigma^2 I_k + L^T L
    logdet()  = logdet C + (n - k) log sigma^2
    sample(t) = t columns drawn from N(0, P):  L g1 + sigma g2
    """

    def __init__(self, kmatmul, n, outputscale, noise, rank, device, dtype=torch.float32, rel_tol=1e-6):
        s, noise = float(outputscale), float(noise)
        rank = int(min(rank, n))
        Lt = torch.zeros(rank, n, dtype=dtype, device=device)           # L^T: every pivot's column is one contiguous row
        diag = torch.full((n,), s, dtype=dtype, device=device)          # s * diag(K), diag(K) = 1
        onehot = torch.zeros(n, 1, dtype=dtype, device=device)
        for m in range(rank):
            piv = torch.argmax(diag).reshape(1)
            dmax = diag[piv]
            onehot.zero_().index_fill_(0, piv, 1.0)
            row = s * kmatmul(onehot).reshape(-1)
            if m > 0:
                row = row - Lt[:m].t() @ Lt[:m].index_select(1, piv).reshape(-1)
            # a pivot whose residual diagonal has fallen to rounding level contributes nothing (column of zeros)
            ok = dmax > rel_tol * s
            col = torch.where(ok, row / dmax.clamp_min(1e-30).sqrt(), torch.zeros_like(row))
            Lt[m] = col
            diag = (diag - col * col).clamp_min(0).index_fill(0, piv, 0.0)
        self.Lt, self.noise, self.n, self.rank = Lt, noise, n, rank
        C = self._gram(Lt.t()).double() + noise * torch.eye(rank, dtype=torch.float64, device=device)
        self._chol = torch.linalg.cholesky(C)

    @property
    def L(self):
        return self.Lt.t()

    def _gram(self, R, splits=256):
        """L^T R, [k, t]: the reduction runs over the n rows, so it is cut into `splits` independent
        partial products (one batched GEMM) instead of one GEMM with a single long inner loop."""
        n, k = self.n, self.rank
        if n < 64 * splits:
            return self.Lt @ R
        chunk = n // splits
        main = chunk * splits
        A = self.Lt[:, :main].reshape(k, splits, chunk).permute(1, 0, 2)         # [splits, k, chunk], a view
        out = torch.bmm(A, R[:main].reshape(splits, chunk, R.shape[1])).sum(0)
        if main < n:
            out = out + self.Lt[:, main:] @ R[main:]
        return out

    def solve(self, R):
        t = torch.cholesky_solve(self._gram(R).double(), self._chol).to(R.dtype)
        return (R - self.Lt.t() @ t) / self.noise

    def logdet(self):
        return 2.0 * self._chol.diagonal().log().sum() + (self.n - self.rank) * math.log(self.noise)

    def sample(self, t, generator=None):
        g1 = torch.randn(self.rank, t, generator=generator, device=self.Lt.device, dtype=self.Lt.dtype)
        g2 = torch.randn(self.n, t, generator=generator, device=self.Lt.device, dtype=self.Lt.dtype)
        return self.Lt.t() @ g1 + math.sqrt(self.noise) * g2


def _vp(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_host_threads_checked = False


def cap_host_threads(force=False):
    """Once per process, before the first small host factorisation of a solve (the k x k Cholesky of the
    preconditioner, the Lanczos tridiagonals' eigen-decompositions): if the process runs under a cgroup CPU quota
    smaller than the thread pool torch / MKL would start (a container that SEES 256 CPUs but may use 16 of them -- the
    MI355X boxes), cap the pool at half the quota.  A 128-thread pool spinning for a 100 x 100 Cholesky uses up the
    quota of the scheduler period and the whole process is throttled for the rest of it: measured as 30-100 ms stalls
    in a few of every hundred calls (cpu.stat: nr_throttled), in whatever the host thread happened to be doing.
    Nothing is changed when OMP_NUM_THREADS / MKL_NUM_THREADS are set (the user has decided) or no quota applies.
    Returns the thread count in force."""
    global _host_threads_checked
    if _host_threads_checked and not force:
        return torch.get_num_threads()
    _host_threads_checked = True
    import os
    if os.environ.get("OMP_NUM_THREADS") or os.environ.get("MKL_NUM_THREADS"):
        return torch.get_num_threads()
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]               # cgroup v2
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())         # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        cap = max(1, int(quota) // 2)
        if torch.get_num_threads() > cap:
            torch.set_num_threads(cap)
    return torch.get_num_threads()


_host_cap = None          # the thread count the small host factorisations run under (None: not worked out yet)


class _small_host_factorisation:
    """`with _small_host_factorisation():` around a k x k host Cholesky / a stack of tridiagonal eigen-problems: the
    BLAS pool is capped (cap_host_threads' rule) for the duration and RESTORED afterwards -- the library never changes
    the host application's thread count for good; a process that wants the cap everywhere calls cap_host_threads()
    itself (bench.py does)."""

    def __enter__(self):
        global _host_cap, _host_threads_checked
        self.before = torch.get_num_threads()
        if _host_cap is None:
            checked = _host_threads_checked
            _host_cap = cap_host_threads(force=True)
            _host_threads_checked = checked
        if _host_cap < self.before:
            torch.set_num_threads(_host_cap)
        elif torch.get_num_threads() != self.before:
            torch.set_num_threads(self.before)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.before:
            torch.set_num_threads(self.before)
        return False


class LatticePreconditioner:
    """The same preconditioner, P = L L^T + sigma^2 I with L the rank-k pivoted Cholesky factor of s K, on the HIP path
    (experiments/train_simplexgp.py:36: the reference trains with max_preconditioner_size(100)).

    What differs from PivotedCholeskyPreconditioner (kept as the CPU / reference form) is where the work runs:
      * everything lives in LATTICE row order (L^T is [kp][ld], its n dimension ordered like the lattice's points), so a
        preconditioned CG iteration permutes nothing;
      * the factor is built in batches of speculated pivots (plx_pchol_*): up to `batch` kernel rows per MVM instead of
        one -- computed on the frontier of their non-zero vertex rows (plx_filter_onehot; sparse_rows) --, the pivots the
        sequential algorithm takes among them decided on the device (planned on the candidates' own block of the panel
        while a bound on the other entries vouches for them, then, if asked, step by step against the true argmax of
        the updated residual diagonal: exact_steps, None = when it pays) -- the factor is the sequential algorithm's,
        pivot for pivot (ties broken by the caller's row number, like torch.argmax on the caller-order diagonal), at
        one host read-back per batch;
      * solve() = plx_pcg_project (L^T R on the matrix cores, C^-1 in fp64) + plx_pcg_apply;
      * the finished factor is kept in fp16 (factor_dtype; fp32 on request): both passes of an application stream the
        factor and nothing else of size, so its width is their time.  The factor is BUILT in fp32 and rounded once;
        C, the log-determinant, solve() and sample() all use the rounded factor, i.e. P is exactly
        L~ L~^T + sigma^2 I for the stored L~ -- a preconditioner has to be SPD and the same matrix everywhere it
        appears in the estimator, nothing more.
    """

    MAX_RANK = 1024          # factor columns the native passes hold (plx_pcg.hip: factor_shape_ok)
    SPARSE_ROWS_MAX_FRACTION = 0.5    # kernel rows are computed on the frontier of their non-zero vertex rows while a batch's frontier stays under this share of the lattice

    def __init__(self, lat, outputscale, noise, rank, rel_tol=1e-6, batch=16, factor_dtype=torch.float16, sparse_rows=True,
                 exact_steps=None):
        import ctypes
        from . import _native as nv
        lib = nv.lib()
        if min(int(rank), lat.n_owned) > self.MAX_RANK:
            raise ValueError(f"LatticePreconditioner holds at most {self.MAX_RANK} factor columns (rank {rank}): use "
                             "PivotedCholeskyPreconditioner")
        s, noise = float(outputscale), float(noise)
        dev = lat.device
        n = lat.n_owned
        if lat.n != n:
            raise ValueError("LatticePreconditioner needs a single-shard lattice")
        k = int(min(rank, n))
        kp = max(16, (k + 15) // 16 * 16)
        ld = (n + 63) // 64 * 64
        self.lat, self.n, self.rank, self.kp, self.ld, self.noise, self.outputscale = lat, n, k, kp, ld, noise, s
        self.build_id = lat.build_id        # a lattice object is recycled by the cache: the factor belongs to THIS build of it
        self.ref = None
        self.ref_key = None                 # LatticeGP._positions_key(x): identity + version of x and of the raw lengthscale
        self.Lt = torch.empty(kp, ld, dtype=torch.float32, device=dev)      # rows < k, columns < n: every entry is written by a step
        if kp > k:
            self.Lt[k:].zero_()
        if ld > n:
            self.Lt[:, n:].zero_()
        diag = torch.full((n,), s, dtype=torch.float32, device=dev)
        row_rank = torch.empty(lat.n, dtype=torch.int32, device=dev)          # caller row of every lattice position (uint32 bits)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            nv.check(lib.plx_copy_point_perm(lat._h, _vp(row_rank), stream), "plx_copy_point_perm")
            work = torch.empty(int(lib.plx_pchol_work_bytes(ld, kp)), dtype=torch.uint8, device=dev)
            cand = torch.empty(16, dtype=torch.int32, device=dev)
            status = torch.zeros(3, dtype=torch.int32, device=dev)    # [pivots the batch accepted, of them planned, vertex rows its filter worked on]
            accepted, frontier = status[0:2], status[2:3]
            scale = torch.tensor([s, 1.0], dtype=torch.float32, device=dev)
            was_lattice = lat.lattice_rows
            lat.set_lattice_row_order(True)
            bufs = {}
            m, B, self.batches, self.sparse_batches, self.planned_batches = 0, max(1, min(int(batch), 16)), 0, 0, 0
            sparse = bool(sparse_rows)
            # one launch per step behind the planned ones (plx_pchol_factor_batch): tried when a batch came out partly
            # planned, kept while those launches add pivots, left alone for a few batches when they did not
            exact, cooldown = bool(exact_steps), 0        # exact_steps: None = as described, True / False = always / never (A/B, tests)
            selected = 0         # candidates the work buffer already holds for the coming batch (selected ahead of the read-back)
            try:
                while m < k:
                    nb = min(B, k - m, n)
                    t = nb if nb == 1 else (nb + 3) // 4 * 4
                    if t not in bufs:
                        bufs[t] = (torch.empty(n, t, dtype=torch.float32, device=dev), lat.new_values(t), lat.new_values(t))
                    rows, vals, scratch = bufs[t]
                    if selected != nb:
                        nv.check(lib.plx_pchol_select(_vp(diag), _vp(row_rank), n, nb, ld, kp, _vp(cand), _vp(work), stream),
                                 "plx_pchol_select")
                    # K e_p for the nb candidates: splat, blur and slice of one-hot columns, on the frontier of their
                    # non-zero vertex rows while that is a small part of the lattice (plx_filter_onehot)
                    lat.filter_onehot(cand, nb, vals, scratch, rows, vd=t, sparse=sparse, frontier=frontier)
                    nv.check(lib.plx_pchol_factor_batch(_vp(self.Lt), ld, kp, m, _vp(rows), t, _vp(scale), _vp(cand), nb, _vp(diag),
                                                        _vp(row_rank), n, float(rel_tol * s), int(exact), _vp(accepted), _vp(work),
                                                        stream),
                             "plx_pchol_factor_batch")
                    # the next batch's candidates, selected while the host waits for this one's counts -- for the batch size
                    # that follows if every candidate is accepted (the selection only depends on the diagonal this batch leaves,
                    # so it also serves whenever the next batch turns out to have that size; any other size selects again)
                    selected = min(B, k - m - nb, n)
                    if selected > 0:
                        nv.check(lib.plx_pchol_select(_vp(diag), _vp(row_rank), n, selected, ld, kp, _vp(cand), _vp(work), stream),
                                 "plx_pchol_select")
                    a, planned, front = status.tolist()        # the one host read-back of the batch
                    if not 1 <= a <= nb:               # (an assert would vanish under python -O and a == 0 would spin forever)
                        raise RuntimeError(f"plx_pchol_factor_batch accepted {a} of {nb} speculated pivots")
                    m += a
                    self.batches += 1
                    self.sparse_batches += int(sparse)
                    self.planned_batches += int(planned == nb)
                    if exact_steps is None:
                        if exact:
                            if a == planned:           # the step launches added nothing
                                exact, cooldown = False, 4
                        elif planned < nb:
                            if cooldown == 0:
                                exact = True
                            cooldown = max(0, cooldown - 1)
                    # on a coarse lattice every kernel row touches most vertices: the dense passes are the cheaper ones there
                    if sparse and front > self.SPARSE_ROWS_MAX_FRACTION * lat.m:
                        sparse = False
                    # speculation depth follows what the lattice accepts: dense kernels (few, strongly coupled points)
                    # end a batch at the first or second pivot, sparse ones take every candidate
                    B = min(int(batch), 16, 2 * a) if a < nb else min(int(batch), 16, max(B, 2 * a))
            finally:
                lat.set_lattice_row_order(was_lattice)
        from . import _native as nvc
        self.factor_type = nvc.FACTOR_F32
        if factor_dtype == torch.float16 and 1e-7 < s < 1e9:          # entries of L are bounded by sqrt(s): inside fp16's range
            half = torch.empty(kp, ld, dtype=torch.float16, device=dev)
            with torch.cuda.device(dev):
                nv.check(lib.plx_pcg_factor_to_half(_vp(self.Lt), ld, kp, _vp(half), stream), "plx_pcg_factor_to_half")
            self._factor, self.factor_type = half, nvc.FACTOR_F16
        else:
            self._factor = self.Lt
        # C = sigma^2 I + L^T L in fp64 (64 partial products over the n dimension, summed in fp64) -- of the STORED factor:
        # fp16 entries multiply exactly in fp32, so the matrix cores' fp16 x fp16 -> fp32 products are the rounded
        # factor's own (half the bytes of converting it back first)
        C32 = None
        if self._factor is not self.Lt:
            Ah = half[:k].reshape(k, 64, ld // 64).permute(1, 0, 2)
            try:
                C32 = torch.bmm(Ah, Ah.transpose(1, 2), out_dtype=torch.float32)
            except (NotImplementedError, RuntimeError, TypeError):         # (a torch without bmm(out_dtype=))
                self.Lt.copy_(half)
        if C32 is None:
            A = self.Lt[:k].reshape(k, 64, ld // 64).permute(1, 0, 2)
            C32 = torch.bmm(A, A.transpose(1, 2))
        C = C32.double().sum(0) + noise * torch.eye(k, dtype=torch.float64, device=dev)
        # the k x k factorisation and inverse on the host: 80 KB each way, against a dozen ~100 us launches of the device solver
        C_host = C.cpu()
        with _small_host_factorisation():
            self._chol = torch.linalg.cholesky(C_host)
            cinv = torch.eye(kp, dtype=torch.float64) / noise
            cinv[:k, :k] = torch.cholesky_inverse(self._chol)
        self._cinv = cinv.contiguous().to(dev)
        self._logdet = float(2.0 * self._chol.diagonal().log().sum()) + (n - k) * math.log(noise)
        self._scale_solve = torch.tensor([1.0, 1.0 / noise], dtype=torch.float32, device=dev)
        self._scale_sample = torch.tensor([math.sqrt(noise), 1.0], dtype=torch.float32, device=dev)
        self._T = torch.zeros(kp, 16, dtype=torch.float32, device=dev)
        self._work = {}
        if self._factor is not self.Lt:
            self.Lt = None                                             # only the fp16 copy stays resident

    @property
    def L(self):
        """[n, k] in the caller's row order (tests; the solver never forms it)."""
        return self.lat.from_lattice_order(self._factor[:self.rank, :self.n].float().t().contiguous())

    def _workspace(self, t):
        from . import _native as nv
        w = self._work.get(t)
        if w is None:
            w = self._work[t] = torch.empty(int(nv.lib().plx_pcg_work_floats(self.n, self.kp, t)), dtype=torch.float32,
                                            device=self._factor.device)
        return w

    def solve_lattice(self, R, out=None, rz=None):
        """Z = P^-1 R for R [n, t] in lattice row order (t <= 16); rz (optional, [t]) receives <R, Z> per column."""
        import ctypes
        from . import _native as nv
        lib = nv.lib()
        assert R.is_cuda and R.dtype == torch.float32 and R.dim() == 2 and R.is_contiguous() and R.shape[0] == self.n
        t = R.shape[1]
        Z = torch.empty_like(R) if out is None else out
        work = self._workspace(t)
        dev = self._factor.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            nv.check(lib.plx_pcg_project(_vp(self._factor), self.factor_type, self.ld, self.kp, _vp(R), self.n, t, _vp(self._cinv),
                                         _vp(self._T), _vp(work), stream), "plx_pcg_project")
            nv.check(lib.plx_pcg_apply(_vp(self._factor), self.factor_type, self.ld, self.kp, self.rank, _vp(R), self.n, t,
                                       _vp(self._T), _vp(self._scale_solve), _vp(Z), _vp(rz), _vp(work), stream), "plx_pcg_apply")
        return Z

    def _columns(self, fn, R):
        """fn over column blocks of at most 16 (the native passes' tile), R in the caller's order."""
        if self.lat.build_id != self.build_id:
            raise RuntimeError("LatticePreconditioner: its lattice has been rebuilt for other positions (lattice-cache eviction); "
                               "build the preconditioner again")
        outs = []
        for c0 in range(0, R.shape[1], 16):
            blk = self.lat.to_lattice_order(R[:, c0:c0 + 16]).contiguous()
            outs.append(self.lat.from_lattice_order(fn(blk)))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 1)

    def solve(self, R):
        """P^-1 R, rows in the caller's order (any number of columns)."""
        return self._columns(self.solve_lattice, R.contiguous())

    def logdet(self):
        return self._logdet

    def sample(self, t, generator=None):
        """t columns drawn from N(0, P), rows in the caller's order: L g1 + sigma g2, the same draws in the same order
        as PivotedCholeskyPreconditioner.sample."""
        import ctypes
        from . import _native as nv
        lib = nv.lib()
        dev = self._factor.device
        g1 = torch.randn(self.rank, t, generator=generator, device=dev, dtype=torch.float32)
        g2 = torch.randn(self.n, t, generator=generator, device=dev, dtype=torch.float32)
        if self.lat.build_id != self.build_id:
            raise RuntimeError("LatticePreconditioner: its lattice has been rebuilt for other positions (lattice-cache eviction); "
                               "build the preconditioner again")
        outs = []
        for c0 in range(0, t, 16):                                       # the native pass works on column tiles of at most 16
            blk = self.lat.to_lattice_order(g2[:, c0:c0 + 16]).contiguous()
            tb = blk.shape[1]
            T = torch.zeros(self.kp, 16, dtype=torch.float32, device=dev)
            T[:self.rank, :tb] = -g1[:, c0:c0 + tb]                      # Z = sigma g2 - L (-g1)
            Z = torch.empty_like(blk)
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                nv.check(lib.plx_pcg_apply(_vp(self._factor), self.factor_type, self.ld, self.kp, self.rank, _vp(blk), self.n, tb,
                                           _vp(T), _vp(self._scale_sample), _vp(Z), None, _vp(self._workspace(tb)), stream),
                         "plx_pcg_apply")
            outs.append(self.lat.from_lattice_order(Z))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 1)


def _batched_pcg_native(matmul, B, precond, max_iter, tol, want_tridiag, check_every, matmul_dot=None, floor=0):
    """Preconditioned batched CG on one GPU, rows in lattice order, scalars on the device: per iteration one MVM (with
    its p^T A p), plx_cg_step_update (alpha = rz / pAp, X, R, |R|^2), the preconditioner (plx_pcg_project +
    plx_pcg_apply: Z = P^-1 R and <R, Z>) and plx_pcg_step_direction (beta = rz' / rz, P = Z + beta P, active)."""
    import ctypes
    from . import _native as nv
    lib = nv.lib()
    n, t = B.shape
    dev = B.device
    X = torch.zeros_like(B)
    R = B.clone().contiguous()
    rz = torch.empty(t, dtype=torch.float32, device=dev)
    rz_new = torch.empty_like(rz)
    Z = precond.solve_lattice(R, rz=rz)
    P = Z.clone()
    rz0 = rz.clone()
    rr = _colsum(R, R)
    b_norm = rr.sqrt().clamp_min(1e-30)
    rr = rr.clone()
    active = torch.ones(t, dtype=torch.float32, device=dev)
    active_next = torch.empty_like(active)
    key = (dev.index, t)
    work = _dot_work.get(key)
    if work is None:
        work = _dot_work[key] = torch.empty(int(lib.plx_coldot_work_floats(t)), dtype=torch.float32, device=dev)
    alphas = torch.zeros(max_iter if want_tridiag else 1, t, dtype=torch.float32, device=dev)
    betas = torch.zeros(max_iter if want_tridiag else 1, t, dtype=torch.float32, device=dev)
    it = 0
    # the iteration without its three stand-alone reductions (pAp, |R|^2, <R, Z>): the partial sums are added up inside the
    # update and the direction kernels (plx_cg_step_update_fused, plx_pcg_step_direction_fused)
    partial_mm = getattr(matmul_dot, "partial", None) if _fuse_cg_steps(n) else None
    fused = partial_mm is not None and int(lib.plx_cg_fused_work_floats(t)) > 0
    if fused:
        fkey = (dev.index, t, "fused")
        fwork = _dot_work.get(fkey)
        if fwork is None:
            fwork = _dot_work[fkey] = torch.empty(int(lib.plx_cg_fused_work_floats(t)), dtype=torch.float32, device=dev)
        pwork = precond._workspace(t)
        rz_part = pwork[int(lib.plx_pcg_rz_partial_offset(precond.kp)):]
        nrz = int(lib.plx_pcg_rz_partial_rows(n, precond.factor_type))
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for it in range(1, max_iter + 1):
            if fused:
                AP, pap_part, tiles = partial_mm(P)
                row = it - 1 if want_tridiag else 0
                nv.check(lib.plx_cg_step_update_fused(_vp(X), _vp(R), _vp(P), _vp(AP), _vp(rz), _vp(pap_part), tiles, _vp(active), n, t,
                                                      _vp(alphas[row]), _vp(fwork), stream), "plx_cg_step_update_fused")
                precond.solve_lattice(R, out=Z, rz=None)              # <R, Z> stays as partial sums in the preconditioner's work buffer
                step_tol = float(tol) if it >= floor else min(float(tol), _FROZEN_BELOW)
                nv.check(lib.plx_pcg_step_direction_fused(_vp(P), _vp(Z), _vp(rz_part), nrz, _vp(fwork), _vp(rz), _vp(active), _vp(b_norm),
                                                          step_tol, n, t, _vp(rz_new), _vp(rr), _vp(betas[row]), _vp(active_next),
                                                          stream), "plx_pcg_step_direction_fused")
                rz, rz_new = rz_new, rz
                active, active_next = active_next, active
                if tol > 0 and (it % check_every == 0 or it == max_iter) and not bool(active.any()):
                    break
                continue
            if matmul_dot is not None:
                AP, pAp = matmul_dot(P)
                pAp = pAp.contiguous()
            else:
                AP = matmul(P)
                AP = AP if AP.is_contiguous() else AP.contiguous()
                pAp = _colsum(P, AP)
            row = it - 1 if want_tridiag else 0
            nv.check(lib.plx_cg_step_update(_vp(X), _vp(R), _vp(P), _vp(AP), _vp(rz), _vp(pAp), _vp(active), n, t, _vp(rr),
                                            _vp(alphas[row]), _vp(work), stream), "plx_cg_step_update")
            precond.solve_lattice(R, out=Z, rz=rz_new)
            step_tol = float(tol) if it >= floor else min(float(tol), _FROZEN_BELOW)
            nv.check(lib.plx_pcg_step_direction(_vp(P), _vp(Z), _vp(rz_new), _vp(rz), _vp(rr), _vp(active), _vp(b_norm), step_tol,
                                                n, t, _vp(betas[row]), _vp(active_next), stream), "plx_pcg_step_direction")
            rz, rz_new = rz_new, rz
            active, active_next = active_next, active
            if tol > 0 and (it % check_every == 0 or it == max_iter) and not bool(active.any()):
                break
    info = {"iterations": it, "residual": (rr.sqrt() / b_norm), "rz0": rz0}
    if want_tridiag:
        info["tridiag"] = _tridiag_from_cg(alphas[:it], betas[:it], B, info)
    return X, info


_FROZEN_BELOW = 1e-10     # a column whose relative residual is below this never moves again (GPyTorch: stop_updating_after)


def _iteration_floor(max_iter, want_tridiag, min_iter, min_tridiag_iter):
    """No column is frozen and the loop does not stop before this many iterations: GPyTorch's linear_cg runs at least
    min(10, max_iter - 1) iterations, and at least max_lanczos_quadrature_iterations (20) when tridiagonals are
    requested -- with the reference's training tolerance cg_tolerance(1.0) (experiments/train_simplexgp.py:34) that
    floor is what gives the solve and the SLQ log-det any accuracy at all."""
    floor = min(min_iter, max(max_iter - 1, 0))
    if want_tridiag:
        floor = max(floor, min(min_tridiag_iter, max(max_iter - 1, 0)))
    return floor


def batched_cg(matmul, B, max_iter=1000, tol=1e-4, reduce=None, want_tridiag=False, check_every=4, precond=None,
               matmul_dot=None, min_iter=10, min_tridiag_iter=20, lattice_rows=False):
    """Solve A X = B for all columns of B at once (A symmetric positive definite,
    known through `matmul`).  Stops when every column's residual norm is below
    `tol` x its right-hand-side norm, or after max_iter iterations.  The stopping
    test reads a device flag, i.e. synchronises with the GPU; it runs every
    `check_every` iterations so that the launch queue stays full (converged
    columns are frozen on the device in every iteration, so the extra iterations
    do not change them).

    `precond` (an object with solve(R) = P^-1 R) switches to preconditioned CG; the
    tridiagonals are then those of P^-1/2 A P^-1/2 started at P^-1/2 B, and
    info["rz0"] holds B^T P^-1 B per column (the quadrature weight).

    `matmul_dot` (optional, single-GPU HIP path): V -> (A V, column-wise <V, A V>) in one call, used
    instead of matmul + a separate dot product.

    `min_iter` / `min_tridiag_iter`: iteration floor as in GPyTorch's linear_cg (see _iteration_floor); before it only
    columns that have converged to rounding level (relative residual < 1e-10) are frozen.

    `lattice_rows`: the rows of B (and what `matmul` takes and returns) are in the row order of precond.lat, a
    LatticePreconditioner -- the native preconditioned iteration runs; any other combination goes through
    precond.solve(), which takes rows in the caller's order.

    Returns (X, info); with want_tridiag, info["tridiag"] holds the per-column
    Lanczos tridiagonals rebuilt from the CG coefficients (mBCG), shape [t, k, k].
    """
    floor = _iteration_floor(max_iter, want_tridiag, min_iter, min_tridiag_iter)
    if lattice_rows:
        # only the native iteration works in the preconditioner's row order: precond.solve() takes caller-order rows, and
        # a silently permuted P^-1 is a different preconditioner from the one behind the probes and logdet(P)
        if not (isinstance(precond, LatticePreconditioner) and reduce is None and _native_ok(B) and B.shape[1] <= 16):
            raise ValueError("batched_cg(lattice_rows=True) needs a LatticePreconditioner, reduce=None and a contiguous fp32 "
                             "CUDA right-hand side of at most 16 columns (got %s, reduce=%s, B %s %s)"
                             % (type(precond).__name__, "set" if reduce is not None else None, tuple(B.shape), B.dtype))
        return _batched_pcg_native(matmul, B, precond, max_iter, tol, want_tridiag, check_every, matmul_dot, floor)
    if precond is not None:
        return _batched_pcg(matmul, B, precond, max_iter, tol, reduce, want_tridiag, check_every, floor)
    if reduce is None and _native_ok(B):
        return _batched_cg_native(matmul, B, max_iter, tol, want_tridiag, check_every, matmul_dot, floor)
    X = torch.zeros_like(B)
    R = B.clone().contiguous()
    P = R.clone()
    rs = _colsum(R, R, reduce)
    b_norm = rs.sqrt().clamp_min(1e-30)
    alphas, betas = [], []
    active = torch.ones_like(rs, dtype=torch.bool)
    it = 0
    for it in range(1, max_iter + 1):
        AP = matmul(P).contiguous()
        pAp = _colsum(P, AP, reduce)
        alpha = torch.where(active, rs / pAp.clamp_min(1e-30), torch.zeros_like(rs))
        rs_new = _cg_update(X, R, P, AP, alpha, reduce)
        beta = torch.where(active, rs_new / rs.clamp_min(1e-30), torch.zeros_like(rs))
        if want_tridiag:
            alphas.append(alpha)
            betas.append(beta)
        _cg_direction(P, R, beta)
        rs = rs_new
        active = active & (rs.sqrt() / b_norm > (tol if it >= floor else min(tol, _FROZEN_BELOW)))
        if tol > 0 and (it % check_every == 0 or it == max_iter) and not bool(active.any()):
            break
    info = {"iterations": it, "residual": (rs.sqrt() / b_norm)}
    if want_tridiag:
        info["tridiag"] = _tridiag_from_cg(alphas, betas, B)
    return X, info


# The column reductions of a CG iteration (<P, AP>, |R|^2, and <R, Z> of a preconditioned one) inside the update / direction
# kernels (plx_cg_step_*_fused: 10 launches per plain iteration instead of 12, 12 per preconditioned one instead of 15): True
# always, False never, "auto" = while the vectors have at most FUSED_CG_MAX_ROWS rows.  Measured interleaved
# (tools/ab_cg_steps_r6.py, 50 iterations at 12 columns, ms fused / stand-alone): N = 2e4, d = 4: 2.03 / 2.28; N = 10,623, d = 18:
# 4.88 / 4.98 (rank-100 preconditioner: 6.72 / 6.88); N = 1e5: 2.82 / 2.81; N = 3e5: 12.33 / 12.35; N = 1e6: 28.57 / 28.48
# (preconditioned 35.97 / 35.81): the fold pays where an iteration is launch-bound and costs 0.3-0.5 % where each workgroup's
# redundant sum over the partials (0.56 MB at N = 1e6) outweighs two 7 us launches.
FUSED_CG_STEPS = "auto"
FUSED_CG_MAX_ROWS = 65536


def _fuse_cg_steps(n):
    return FUSED_CG_STEPS is True or (FUSED_CG_STEPS == "auto" and n <= FUSED_CG_MAX_ROWS)


def _batched_cg_native(matmul, B, max_iter, tol, want_tridiag, check_every, matmul_dot=None, floor=0):
    """batched_cg on one GPU with the iteration's scalars kept on the device: per iteration one
    MVM, one column dot, plx_cg_step_update and plx_cg_step_direction (alpha, beta and the
    active mask are formed inside those kernels)."""
    import ctypes
    from . import _native as nv
    lib = nv.lib()
    n, t = B.shape
    dev = B.device
    X = torch.zeros_like(B)
    R = B.clone().contiguous()
    P = R.clone()
    rs = _colsum(R, R)
    b_norm = rs.sqrt().clamp_min(1e-30)
    active = torch.ones(t, dtype=torch.float32, device=dev)
    active_next = torch.empty_like(active)
    rs_new = torch.empty_like(rs)
    key = (dev.index, t)
    work = _dot_work.get(key)
    if work is None:
        work = _dot_work[key] = torch.empty(int(lib.plx_coldot_work_floats(t)), dtype=torch.float32, device=dev)
    kmax = max_iter
    alphas = torch.zeros(kmax if want_tridiag else 1, t, dtype=torch.float32, device=dev)
    betas = torch.zeros(kmax if want_tridiag else 1, t, dtype=torch.float32, device=dev)
    p = lambda a: ctypes.c_void_p(a.data_ptr())          # noqa: E731
    it = 0
    # the iteration without its two stand-alone reductions (plx_cg_step_*_fused): rows of whole 16-byte chunks, and an
    # MVM that can leave its <P, AP> partial sums un-reduced (matmul_dot.partial)
    partial_mm = getattr(matmul_dot, "partial", None) if _fuse_cg_steps(n) else None
    fused = partial_mm is not None and int(lib.plx_cg_fused_work_floats(t)) > 0
    if fused:
        fkey = (dev.index, t, "fused")
        fwork = _dot_work.get(fkey)
        if fwork is None:
            fwork = _dot_work[fkey] = torch.empty(int(lib.plx_cg_fused_work_floats(t)), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for it in range(1, max_iter + 1):
            if fused:
                AP, pap_part, tiles = partial_mm(P)
                row = it - 1 if want_tridiag else 0
                nv.check(lib.plx_cg_step_update_fused(p(X), p(R), p(P), p(AP), p(rs), p(pap_part), tiles, p(active), n, t,
                                                      p(alphas[row]), p(fwork), stream), "plx_cg_step_update_fused")
                step_tol = float(tol) if it >= floor else min(float(tol), _FROZEN_BELOW)
                nv.check(lib.plx_cg_step_direction_fused(p(P), p(R), p(fwork), p(rs), p(active), p(b_norm), step_tol, n, t,
                                                         p(rs_new), p(betas[row]), p(active_next), stream),
                         "plx_cg_step_direction_fused")
                rs, rs_new = rs_new, rs
                active, active_next = active_next, active
                if tol > 0 and (it % check_every == 0 or it == max_iter) and not bool(active.any()):
                    break
                continue
            if matmul_dot is not None:
                AP, pAp = matmul_dot(P)
                pAp = pAp.contiguous()
            else:
                AP = matmul(P)
                AP = AP if AP.is_contiguous() else AP.contiguous()
                pAp = _colsum(P, AP)
            row = it - 1 if want_tridiag else 0
            nv.check(lib.plx_cg_step_update(p(X), p(R), p(P), p(AP), p(rs), p(pAp), p(active), n, t, p(rs_new),
                                            p(alphas[row]), p(work), stream), "plx_cg_step_update")
            step_tol = float(tol) if it >= floor else min(float(tol), _FROZEN_BELOW)
            nv.check(lib.plx_cg_step_direction(p(P), p(R), p(rs_new), p(rs), p(active), p(b_norm), step_tol, n, t,
                                               p(betas[row]), p(active_next), stream), "plx_cg_step_direction")
            rs, rs_new = rs_new, rs
            active, active_next = active_next, active
            if tol > 0 and (it % check_every == 0 or it == max_iter) and not bool(active.any()):
                break
    info = {"iterations": it, "residual": (rs.sqrt() / b_norm)}
    if want_tridiag:
        info["tridiag"] = _tridiag_from_cg(alphas[:it], betas[:it], B, info)
    return X, info


def _tridiag_from_cg(alphas, betas, B, info=None):
    """[t, k, k] Lanczos tridiagonals from the CG coefficients of k iterations (mBCG): T[i, i] = 1 / a_i + b_{i-1} / a_{i-1},
    T[i, i+1] = sqrt(b_i) / a_i; a column that converged early has alpha = 0 afterwards and its tridiagonal is frozen there
    (identity rows).  Whole-array expressions: a loop over the k iterations was 4 k small launches (0.4 ms at k = 20).
    Coefficients that live on the GPU as two [k, t] arrays (the native solves) are copied to the host in ONE transfer and
    the same expressions run there in numpy: fifteen launches of kilobyte-sized kernels kept the GPU idle for 0.2 ms while
    the host issued them (and the same fifteen expressions as torch CPU ops cost as much in dispatch), and the quadrature
    that consumes the tridiagonals runs on the host anyway -- `info` (optional) receives the host copy as
    info["tridiag_host"]."""
    if torch.is_tensor(alphas) and alphas.is_cuda:
        import numpy as np
        dev = alphas.device
        ab = torch.stack([alphas, betas], 0).cpu().numpy().astype(np.float64)          # [2, k, t]: one copy, one synchronisation
        a, b = ab[0], ab[1]
        k, t = a.shape
        valid = a > 0
        inv_a = np.where(valid, 1.0 / np.maximum(a, 1e-300), 0.0)
        diag = inv_a.copy()
        diag[1:] += b[:-1] * inv_a[:-1]
        diag = np.where(valid, diag, 1.0)
        Tn = np.zeros((t, k, k))
        idx = np.arange(k)
        Tn[:, idx, idx] = diag.T
        if k > 1:
            off = np.where(valid[1:], np.sqrt(np.maximum(b[:-1], 0.0)) * inv_a[:-1], 0.0)      # [k - 1, t]
            Tn[:, idx[:-1], idx[1:]] = off.T
            Tn[:, idx[1:], idx[:-1]] = off.T
        T = torch.from_numpy(Tn)
        if info is not None:
            info["tridiag_host"] = T
        return T.to(dev)
    a = (alphas if torch.is_tensor(alphas) else torch.stack(alphas, 0)).double()           # [k, t]
    b = (betas if torch.is_tensor(betas) else torch.stack(betas, 0)).double()
    k = a.shape[0]
    valid = a > 0
    inv_a = torch.where(valid, 1.0 / a.clamp_min(1e-300), torch.zeros_like(a))
    diag = inv_a.clone()
    diag[1:] += b[:-1] * inv_a[:-1]
    diag = torch.where(valid, diag, torch.ones_like(diag))
    T = torch.diag_embed(diag.t())                # [t, k, k]
    if k > 1:
        off = torch.where(valid[1:], b[:-1].clamp_min(0).sqrt() * inv_a[:-1], torch.zeros_like(inv_a[:-1]))      # [k - 1, t]
        T = T + torch.diag_embed(off.t(), offset=1) + torch.diag_embed(off.t(), offset=-1)
    return T


def _batched_pcg(matmul, B, precond, max_iter, tol, reduce, want_tridiag, check_every, floor=0):
    X = torch.zeros_like(B)
    R = B.clone().contiguous()
    Z = precond.solve(R).contiguous()
    P = Z.clone()
    rz = _colsum(R, Z, reduce)
    rz0 = rz.clone()
    b_norm = _colsum(R, R, reduce).sqrt().clamp_min(1e-30)
    alphas, betas = [], []
    active = torch.ones_like(rz, dtype=torch.bool)
    it = 0
    rr = b_norm ** 2
    for it in range(1, max_iter + 1):
        AP = matmul(P).contiguous()
        pAp = _colsum(P, AP, reduce)
        alpha = torch.where(active, rz / pAp.clamp_min(1e-30), torch.zeros_like(rz))
        rr = _cg_update(X, R, P, AP, alpha, reduce)
        Z = precond.solve(R).contiguous()
        rz_new = _colsum(R, Z, reduce)
        beta = torch.where(active, rz_new / rz.clamp_min(1e-30), torch.zeros_like(rz))
        if want_tridiag:
            alphas.append(alpha)
            betas.append(beta)
        _cg_direction(P, Z, beta)
        rz = rz_new
        active = active & (rr.sqrt() / b_norm > (tol if it >= floor else min(tol, _FROZEN_BELOW)))
        if tol > 0 and (it % check_every == 0 or it == max_iter) and not bool(active.any()):
            break
    info = {"iterations": it, "residual": (rr.sqrt() / b_norm), "rz0": rz0}
    if want_tridiag:
        info["tridiag"] = _tridiag_from_cg(alphas, betas, B)
    return X, info


def slq_terms(tridiag):
    """e1^T log(T_i) e1 per column of a [t, k, k] stack of Lanczos tridiagonals (the quadrature of slq_logdet).  The
    eigen-decompositions of a few dozen k x k matrices run on the HOST (the device solver takes ~1 ms in a dozen small
    launches for eleven 20 x 20 problems; the copy is a few KB and the solve has just synchronised anyway); a tridiagonal
    stack that already is a host tensor (info["tridiag_host"] of the native solves) gives host terms."""
    dev = tridiag.device
    small = tridiag.shape[0] * tridiag.shape[-1] ** 2 <= 1 << 18
    if small and not tridiag.requires_grad:                        # (a tridiagonal that carries a graph keeps the torch path)
        host = tridiag.detach().cpu() if dev.type == "cuda" else tridiag.detach()
        with _small_host_factorisation():
            evals, evecs = torch.linalg.eigh(host)
        ev, first = evals.numpy(), evecs[:, 0, :].numpy()          # (numpy for the rest: a dozen microsecond-sized expressions)
        import numpy as np
        terms = torch.from_numpy((first * first * np.log(np.maximum(ev, 1e-30))).sum(-1))
        return terms.to(dev) if dev.type == "cuda" else terms
    evals, evecs = torch.linalg.eigh(tridiag)
    return ((evecs[:, 0, :] ** 2) * evals.clamp_min(1e-30).log()).sum(-1)


def slq_logdet(tridiag, n, weights=None):
    """Stochastic Lanczos quadrature: logdet(A) ~ mean_i |z_i|^2 e1^T log(T_i) e1; Rademacher
    probes have |z|^2 = n, preconditioned probes pass their own weights (b^T P^-1 b)."""
    quad = slq_terms(tridiag)
    if weights is None:
        return float(n) * quad.mean()
    return (weights.double() * quad.to(weights.device)).mean()


class LatticeGP(nn.Module):
    """Constant mean + outputscale * LatticeKernel + Gaussian noise >= min_noise:
    the model of tests/train_snelson.py:11-23 / experiments/train_simplexgp.py:13-26."""

    def __init__(self, kernel, min_noise=1e-4):
        super().__init__()
        assert isinstance(kernel, LatticeAccelerated)
        self.kernel = kernel
        self.mean = nn.Parameter(torch.zeros(()))
        self.raw_outputscale = nn.Parameter(torch.zeros(()))
        self.raw_noise = nn.Parameter(torch.zeros(()))
        self.min_noise = min_noise

    @property
    def outputscale(self):
        return F.softplus(self.raw_outputscale)

    @property
    def noise(self):
        return F.softplus(self.raw_noise) + self.min_noise

    def khat_matmul(self, x, K=None):
        """V -> (s K(x,x) + sigma^2 I) V as a differentiable closure (K: an already built kernel(x, x))."""
        K = self.kernel(x, x) if K is None else K

        def mm(V):
            return self.outputscale * K.matmul(V) + self.noise * V
        return mm

    @contextlib.contextmanager
    def khat_in_lattice_rows(self, x, K=None):
        """with ... as (mm, to_rows, from_rows): V -> (s K + sigma^2 I) V on vectors whose rows are in the lattice's own
        point order (no gradients), and the two permutations between that order and the caller's.  What khat_solve runs
        its iteration on, offered to other Krylov loops (training.lanczos): every MVM is splat -> blur -> slice + the affine
        tail in ONE pass over the rows, without the two row permutations of the caller-order MVM.  Away from the HIP
        path: the ordinary closure and identity permutations."""
        from . import lattice_kernel as lk
        with torch.no_grad():
            if lk.LatticeFilterGeneral.method is not None or not x.is_cuda:
                yield self.khat_matmul(x, K), (lambda v: v), (lambda v: v)
                return
            ref = lk.carry_hint(K.x.detach(), K.x) if isinstance(K, lk.SquareLazyLattice) \
                else lk.position_hint(x.div(self.kernel.lengthscale), x, scale_of=getattr(self.kernel, "raw_lengthscale", None))
            ref = ref if ref.is_contiguous() else lk.carry_hint(ref.contiguous(), ref)
            lat = lk.lattice_cache().get(ref, self.kernel.dkernel_fn.get_coeffs())
            ss = torch.stack([self.outputscale.detach().reshape(()), self.noise.detach().reshape(())]).to(torch.float32).contiguous()
            lat.set_lattice_row_order(True)
            try:
                yield (lambda V: lat.apply_affine(V.contiguous(), ss)), lat.to_lattice_order, lat.from_lattice_order
            finally:
                lat.set_lattice_row_order(False)

    def _positions_key(self, x):
        """What x / lengthscale was derived from: x by identity (a weak reference: a freed tensor whose address the
        caching allocator hands to another one cannot match) and version, the raw lengthscale PARAMETER by identity and
        version (kernel.lengthscale is a fresh softplus output on every access: its version counter says nothing)."""
        import weakref
        raw = getattr(self.kernel, "raw_lengthscale", None)
        return (weakref.ref(x), x._version, tuple(x.shape), None if raw is None else weakref.ref(raw),
                None if raw is None else raw._version)

    def _hyper_snapshot(self):
        """The values every operator of this model is a function of (raw lengthscale, raw outputscale, raw noise), cloned:
        what a remembered lattice / preconditioner is checked against.  Version counters cannot vouch for parameters --
        torch.optim.Adam(fused=True) writes them without moving the counter (measured on this image) -- so 'nothing
        moved' is confirmed by comparing these few floats on the device (one read-back, only after the cheap identity and
        version checks have passed)."""
        raw = getattr(self.kernel, "raw_lengthscale", None)
        parts = ([] if raw is None else [raw.detach().reshape(-1)]) + [self.raw_outputscale.detach().reshape(1),
                                                                     self.raw_noise.detach().reshape(1)]
        return torch.cat([p.to(torch.float32) for p in parts])

    def _same_hyper(self, pre):
        snap = getattr(pre, "hyper_snapshot", None)
        if snap is None or getattr(pre, "min_noise", None) != float(self.min_noise):
            return False
        now = self._hyper_snapshot()
        return snap.shape == now.shape and snap.device == now.device and bool(torch.equal(snap, now))

    def _same_positions(self, pre, x):
        """pre was built from this x (identity, version, shape) under the lengthscale the kernel has NOW (the parameter's
        identity and version first, then its values against the copy taken at the build)."""
        key = pre.ref_key
        if key is None or len(key) != 5:
            return False
        raw = getattr(self.kernel, "raw_lengthscale", None)
        xr, xv, shape, rr, rv = key
        if xr() is not x or xv != x._version or shape != tuple(x.shape):
            return False
        if raw is None:
            return rr is None
        if not (rr is not None and rr() is raw and rv == raw._version):
            return False
        snap = getattr(pre, "hyper_snapshot", None)
        if snap is None:
            return True
        k = raw.numel()
        return snap.numel() == k + 2 and bool(torch.equal(snap[:k], raw.detach().reshape(-1).to(torch.float32)))

    def __getstate__(self):
        # (copy.deepcopy / pickling of the module: the remembered preconditioner holds device handles of this process)
        state = self.__dict__.copy()
        state.pop("_last_preconditioner", None)
        return state

    def khat_solve(self, x, rhs, K=None, **cg_args):
        """(s K + sigma^2 I)^-1 rhs by batched CG, no gradients.  On the HIP path the
        iteration runs in lattice row order: the right-hand side is permuted once, every
        MVM skips its two row permutations, the solution is permuted back once (dot
        products do not care about row order)."""
        from . import lattice_kernel as lk
        with torch.no_grad():
            if lk.LatticeFilterGeneral.method is not None or not x.is_cuda:
                return batched_cg(self.khat_matmul(x, K), rhs, **cg_args)
            # the positions of an existing operator share storage with its tensor: same lattice-cache key, so the
            # differentiable MVM that follows a solve reuses the lattice built here
            ref = lk.carry_hint(K.x.detach(), K.x) if isinstance(K, lk.SquareLazyLattice) \
                else lk.position_hint(x.div(self.kernel.lengthscale), x, scale_of=getattr(self.kernel, "raw_lengthscale", None))
            ref = ref if ref.is_contiguous() else lk.carry_hint(ref.contiguous(), ref)
            pre = cg_args.get("precond")
            if isinstance(pre, LatticePreconditioner) and pre.ref is not None and pre.ref is not ref \
                    and (pre.ref.data_ptr(), pre.ref._version, pre.ref.shape) != (ref.data_ptr(), ref._version, ref.shape) \
                    and self._same_positions(pre, x):
                # the same x and lengthscale the preconditioner was built from, as a fresh tensor (no K handed over): solve
                # on the preconditioner's lattice.  (Decided from the identities and version counters of x ITSELF -- held
                # through a weak reference, a recycled address cannot pass for it -- and of the raw lengthscale PARAMETER,
                # then from the parameter's d values against the copy taken when the factor was built: one small
                # read-back, and only on this path -- positions that share the factor's storage are not asked.
                # Comparing the n x d positions on the device cost a pass and a host synchronisation per solve.)
                ref = pre.ref
            lat = lk.lattice_cache().get(ref, self.kernel.dkernel_fn.get_coeffs())
            s, noise = self.outputscale, self.noise
            lat.set_lattice_row_order(True)
            try:
                native_pre = isinstance(pre, LatticePreconditioner) and pre.lat is lat and pre.build_id == lat.build_id
                if pre is not None and not (native_pre and rhs.shape[1] <= 16):
                    # a factor whose rows are in the caller's order: keep that order for the whole solve
                    lat.set_lattice_row_order(False)
                    return batched_cg(lambda V: lat.apply(V).mul_(s).addcmul_(V, noise), rhs, **cg_args)
                ss = torch.stack([s.detach().reshape(()), noise.detach().reshape(())]).to(torch.float32).contiguous()
                # columns padded to a multiple of 4 with zero right-hand sides: rows become whole 16-byte vectors, so
                # splat reads the CG vectors in place (no padding copy) and slice writes 16 bytes per lane; a zero
                # column converges at once (alpha = beta = 0) and is dropped from the results
                t = rhs.shape[1]
                pad = (-t) % 4 if t > 1 else 0
                rhs_l = lat.to_lattice_order(rhs)
                if pad:
                    rhs_l = torch.cat([rhs_l, rhs_l.new_zeros(rhs_l.shape[0], pad)], 1).contiguous()
                fused_dot = None
                if 2 <= rhs_l.shape[1] <= 256:
                    def fused_dot(V):
                        return lat.apply_affine(V, ss, want_dot=True)
                    fused_dot.partial = lambda V: lat.apply_affine(V, ss, want_dot="partial")
                sol, info = batched_cg(lambda V: lat.apply_affine(V, ss), rhs_l, matmul_dot=fused_dot, lattice_rows=native_pre,
                                       **cg_args)
                if pad:
                    sol = sol[:, :t].contiguous()
                    info = dict(info, residual=info["residual"][:t])
                    if "tridiag" in info:
                        info["tridiag"] = info["tridiag"][:t]
                    if "tridiag_host" in info:
                        info["tridiag_host"] = info["tridiag_host"][:t]
                    if "rz0" in info:
                        info["rz0"] = info["rz0"][:t]
            finally:
                lat.set_lattice_row_order(False)
            return lat.from_lattice_order(sol), info

    def preconditioner(self, x, rank, K=None, factor_dtype=torch.float16):
        """Rank-`rank` pivoted-Cholesky preconditioner of (s K + sigma^2 I) (no gradients).  On the HIP path it is
        built and applied natively, in the row order of the lattice the solve runs on (LatticePreconditioner)."""
        from . import lattice_kernel as lk
        with torch.no_grad():
            # (the native passes hold at most LatticePreconditioner.MAX_RANK factor columns: larger ranks take the torch form)
            if lk.LatticeFilterGeneral.method is None and x.is_cuda and x.dtype == torch.float32 \
                    and min(int(rank), x.shape[0]) <= LatticePreconditioner.MAX_RANK:
                ref = lk.carry_hint(K.x.detach(), K.x) if isinstance(K, lk.SquareLazyLattice) \
                    else lk.position_hint(x.div(self.kernel.lengthscale), x, scale_of=getattr(self.kernel, "raw_lengthscale", None))
                ref = ref if ref.is_contiguous() else lk.carry_hint(ref.contiguous(), ref)
                lat = lk.lattice_cache().get(ref, self.kernel.dkernel_fn.get_coeffs())
                # the factor of the same operator again (an evaluation, then the next training step: the optimiser has not
                # moved between them): the one built last, if nothing it depends on has been written since
                last = self.__dict__.get("_last_preconditioner")
                if last is not None and last.lat is lat and last.build_id == lat.build_id and last.asked == (int(rank), factor_dtype) \
                        and self._same_positions(last, x) and self._same_hyper(last):
                    self.preconditioner_reuses = self.__dict__.get("preconditioner_reuses", 0) + 1
                    return last
                pre = LatticePreconditioner(lat, self.outputscale, self.noise, rank, factor_dtype=factor_dtype)
                pre.ref = ref          # the positions its lattice was built on (kept alive: the lattice-cache key)
                pre.ref_key = self._positions_key(x)
                pre.asked = (int(rank), factor_dtype)
                pre.hyper_snapshot, pre.min_noise = self._hyper_snapshot(), float(self.min_noise)
                self.__dict__["_last_preconditioner"] = pre
                return pre
            K = self.kernel(x, x) if K is None else K
            return PivotedCholeskyPreconditioner(K.matmul, x.shape[0], self.outputscale, self.noise, rank,
                                                 device=x.device, dtype=x.dtype)


class _Phases:
    """Wall time per phase of one marginal-likelihood evaluation (profile= of marginal_log_likelihood): every mark
    synchronises the device, so it is a measuring mode, never the default."""

    def __init__(self, sink, device):
        import time
        self.sink, self.device, self._clock = sink, device, time.perf_counter
        self._t = self._now() if sink is not None else None

    def _now(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return self._clock()

    def mark(self, name):
        if self.sink is not None:
            t = self._now()
            self.sink[name] = self.sink.get(name, 0.0) + (t - self._t) * 1e3
            self._t = t


def marginal_log_likelihood(model, x, y, num_probes=10, max_cg_iter=1000, cg_tol=1e-4, seed=0, pre_size=0, reduce=None,
                            n_total=None, profile=None):
    """Per-datapoint log marginal likelihood (the quantity GPyTorch's
    ExactMarginalLogLikelihood returns) of a LatticeGP, differentiable with
    respect to every hyper-parameter.

    Value: CG solve for K^-1 (y - mean) + SLQ log-det from the same CG run over
    Rademacher probes.  Gradient: the usual surrogate
        S = -u^T r + 1/2 u^T K u - 1/(2t) sum_i w_i^T K z_i,   u = K^-1 r, w_i = K^-1 z_i (detached)
    whose gradient equals that of the MLL; it costs one more MVM (vd = 1 + t)
    and its backward (one wide filter, py:113-122).

    pre_size > 0 (GPyTorch's max_preconditioner_size; the reference trains with 100)
    preconditions the solve with P = pivoted-Cholesky(s K, pre_size) + sigma^2 I:
    probes are drawn from N(0, P), logdet = logdet P + SLQ of P^-1/2 (sK + sigma^2 I) P^-1/2,
    and the log-det gradient pairs w_i with P^-1 z_i  (E[P^-1 z z^T] = I).

    profile: a dict that receives the wall time in ms of every phase (preconditioner, probes, solve, slq,
    mvm_forward; the caller times backward()); each phase boundary synchronises the device.
    """
    if reduce is not None or n_total is not None:
        # kept in the signature for one release: the old row-sharded form evaluated a block-diagonal likelihood
        raise NotImplementedError("marginal_log_likelihood is single-process; for a row-sharded solve use "
                                  "simplex_gp_amd.distributed.sharded_solve (the reduce= / n_total= arguments were removed)")
    # Single-process only: the kernel operator below is built from `x` alone.  A row-sharded job has to go through
    # distributed.ShardedLatticeMVM (whose vertex all-reduce couples the ranks); distributed.sharded_solve does the
    # solve that way, a sharded marginal likelihood with gradients is not implemented.
    n = n_local = y.shape[0]
    ph = _Phases(profile, y.device)
    r = (y - model.mean).reshape(-1, 1)
    g = torch.Generator(device=y.device).manual_seed(seed)      # on the device: 1e7 CPU draws cost ~0.1 s per step
    K = model.kernel(x, x)
    mm = model.khat_matmul(x, K)
    precond = None
    if pre_size > 0:
        precond = model.preconditioner(x, pre_size, K=K)
    ph.mark("preconditioner")
    with torch.no_grad():
        if precond is None:
            Z = torch.randint(0, 2, (n_local, num_probes), generator=g, device=y.device).to(r.dtype) * 2 - 1
        else:
            Z = precond.sample(num_probes, generator=g)
        rhs = torch.cat([r.detach(), Z], 1)
        ph.mark("probes")
        sol, info = model.khat_solve(x, rhs, K=K, max_iter=max_cg_iter, tol=cg_tol, want_tridiag=True, precond=precond)
        ph.mark("solve")
        u, W = sol[:, :1], sol[:, 1:]
        quad = _colsum(r.detach(), u).sum()
        if precond is not None:
            Z = precond.solve(Z)
    # the differentiable MVM is enqueued BEFORE the host-side quadrature (eigen-decompositions of the Lanczos tridiagonals,
    # a few hundred microseconds of LAPACK): the GPU works on the one while the host does the other
    KV = mm(torch.cat([u, Z], 1))
    s_quad = -(u * r).sum() + 0.5 * (u * KV[:, :1]).sum()
    s_logdet = -0.5 * (W * KV[:, 1:]).sum() / num_probes
    surrogate = s_quad + s_logdet
    ph.mark("mvm_forward")
    with torch.no_grad():
        tri = info.get("tridiag_host", info["tridiag"])
        if precond is None:
            logdet = slq_logdet(tri[1:], n)
        else:
            logdet = precond.logdet() + slq_logdet(tri[1:], n, weights=info["rz0"][1:])
        logdet = logdet.to(quad.device) if torch.is_tensor(logdet) else logdet
        value = -0.5 * quad - 0.5 * logdet - 0.5 * n * math.log(2 * math.pi)
    ph.mark("slq")
    out = (value.to(surrogate.dtype) + (surrogate - surrogate.detach())) / n
    out.cg_info = info
    return out


# ----------------------------------------------------------------------------
# dense exact GP (the ExactModel of tests/train_snelson.py:26-46), for config 1

class ExactRBFGP(nn.Module):
    def __init__(self, min_noise=1e-4):
        super().__init__()
        self.mean = nn.Parameter(torch.zeros(()))
        self.raw_outputscale = nn.Parameter(torch.zeros(()))
        self.raw_lengthscale = nn.Parameter(torch.zeros(()))
        self.raw_noise = nn.Parameter(torch.zeros(()))
        self.min_noise = min_noise

    def mll(self, x, y):
        """Per-datapoint exact log marginal likelihood with K = s exp(-|x-x'|^2 / (2 l^2)) + sigma^2 I."""
        n = y.shape[0]
        ell = F.softplus(self.raw_lengthscale)
        d2 = torch.cdist(x / ell, x / ell).pow(2)
        K = F.softplus(self.raw_outputscale) * torch.exp(-0.5 * d2)
        K = K + (F.softplus(self.raw_noise) + self.min_noise) * torch.eye(n, dtype=x.dtype, device=x.device)
        Lc = torch.linalg.cholesky(K)
        r = (y - self.mean).reshape(-1, 1)
        alpha = torch.cholesky_solve(r, Lc)
        return (-0.5 * (r * alpha).sum() - Lc.diagonal().log().sum() - 0.5 * n * math.log(2 * math.pi)) / n
