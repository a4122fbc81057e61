"""The sharded MVM choreography under gloo, world_size 2 and 3, on CPU.

The lattice engine is the oracle-backed adapter (tests/oracle_lattice_adapter.py);
what is under test is simplex_gp_amd.distributed: row ownership, the single
all-reduce of the vertex accumulators, replicated blur, row-sharded output.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simplex_gp_amd.distributed import ShardedLatticeMVM, shard_bounds

RBF1 = np.array([0.34608543, 1.0, 0.34608543], np.float32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, vd, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.oracle_lattice_adapter import OracleLattice
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        x = torch.randn(n, d, generator=g)
        v = torch.randn(n, vd, generator=g)
        op = ShardedLatticeMVM(x, RBF1, lattice=OracleLattice())
        lo, hi = shard_bounds(n, world, rank)
        assert (op.lo, op.hi) == (lo, hi)
        out_local = op.matmul(v[lo:hi])
        assert out_local.shape == (hi - lo, vd)
        full = op.gather_rows(out_local)
        # 1-D right-hand side takes the squeeze path
        out1 = op.matmul(v[lo:hi, 0])
        assert out1.shape == (hi - lo,) and torch.allclose(out1, out_local[:, 0])
        with pytest.raises(ValueError):
            op.matmul(v[: hi - lo + 1])
        if rank == 0:
            np.save(os.path.join(outdir, "out.npy"), full.numpy())
            np.save(os.path.join(outdir, "m.npy"), np.array(op.m))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_mvm_matches_single_process(tmp_path, world):
    from oracle import oracle
    n, d, vd = 1001, 3, 2          # 1001 rows: uneven shards
    mp.spawn(_worker, args=(world, _free_port(), n, d, vd, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, d, generator=g)
    v = torch.randn(n, vd, generator=g)
    oracle.set_exact_mode(False)
    want, m = oracle.filter(v.numpy(), x.numpy(), RBF1, return_m=True)
    oracle.set_exact_mode(True)
    got = np.load(tmp_path / "out.npy")
    assert int(np.load(tmp_path / "m.npy")) == m
    assert got.shape == want.shape
    # partial splats are summed in a different order than the single-process splat
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-6


def _local_rows_worker(rank, world, port, n, d, vd, outdir):
    """The path bench.py --gpus N takes: every rank holds only its rows, local build -> all-gather of the vertex
    keys -> merge -> sharded MVM; then a sharded CG solve on top of it."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.oracle_lattice_adapter import OracleLattice
    from simplex_gp_amd.distributed import sharded_solve
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(12)
        x = torch.randn(n, d, generator=g)
        v = torch.randn(n, vd, generator=g)
        lo, hi = shard_bounds(n, world, rank)
        op = ShardedLatticeMVM.from_local_rows(x[lo:hi].contiguous(), RBF1, lattice=OracleLattice(), n_total=n)
        assert (op.lo, op.hi, op.n) == (lo, hi, n)
        assert op.key_bytes_exchanged > 0 and op.exchange_bytes(vd) == op.m * vd * 4
        out_local = op.matmul(v[lo:hi])
        full = op.gather_rows(out_local)
        # rebuilding on the same object (new positions) keeps working: what a training loop / bench cadence does
        op.rebuild((x[lo:hi] * 1.5).contiguous(), RBF1)
        full2 = op.gather_rows(op.matmul(v[lo:hi]))
        op.rebuild(x[lo:hi].contiguous(), RBF1)
        sol, info = sharded_solve(op, v[lo:hi].contiguous(), outputscale=0.7, noise=0.3, max_iter=200, tol=1e-6)
        sol_full = op.gather_rows(sol)
        if rank == 0:
            np.save(os.path.join(outdir, "out.npy"), full.numpy())
            np.save(os.path.join(outdir, "out2.npy"), full2.numpy())
            np.save(os.path.join(outdir, "sol.npy"), sol_full.numpy())
            np.save(os.path.join(outdir, "m.npy"), np.array(op.m))
            np.save(os.path.join(outdir, "res.npy"), info["residual"].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build_from_local_rows_matches_oracle(tmp_path, world):
    from oracle import oracle
    n, d, vd = 1001, 3, 2
    mp.spawn(_local_rows_worker, args=(world, _free_port(), n, d, vd, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, d, generator=g)
    v = torch.randn(n, vd, generator=g)
    oracle.set_exact_mode(False)
    try:
        want, m = oracle.filter(v.numpy(), x.numpy(), RBF1, return_m=True)
        want2 = oracle.filter(v.numpy(), (x * 1.5).numpy(), RBF1)
        assert int(np.load(tmp_path / "m.npy")) == m                      # the merged vertex set IS the full lattice's
        for got, ref in ((np.load(tmp_path / "out.npy"), want), (np.load(tmp_path / "out2.npy"), want2)):
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-6
        # the sharded solve satisfies the single-process operator equation
        sol = np.load(tmp_path / "sol.npy")
        back = 0.7 * oracle.filter(sol, x.numpy(), RBF1) + 0.3 * sol
        assert np.linalg.norm(back - v.numpy()) / np.linalg.norm(v.numpy()) <= 1e-4
        assert float(np.load(tmp_path / "res.npy").max()) <= 1e-6
    finally:
        oracle.set_exact_mode(True)


def _forced_world1_worker(rank, world, port, n, d, vd, outdir):
    """distributed.FORCE_COLLECTIVES with ONE rank (the switch behind tests/checks/rccl_world1.py and bench.py's
    single-rank RCCL rehearsal), here under gloo: the build takes the local-rows path (key all-gather, merge) and every
    MVM runs its all-reduce; the numbers equal those of the plain path."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.oracle_lattice_adapter import OracleLattice
    from simplex_gp_amd import distributed as pd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(13)
        x = torch.randn(n, d, generator=g)
        v = torch.randn(n, vd, generator=g)
        plain = ShardedLatticeMVM.from_local_rows(x, RBF1, lattice=OracleLattice(), n_total=n)
        assert not hasattr(plain, "key_bytes_exchanged")                # one rank, nothing forced: no exchange at all
        want = plain.matmul(v)
        pd.FORCE_COLLECTIVES = True
        try:
            op = ShardedLatticeMVM.from_local_rows(x, RBF1, lattice=OracleLattice(), n_total=n)
            assert op.key_bytes_exchanged > 0 and (op.lo, op.hi, op.n) == (0, n, n) and op.m == plain.m
            got = op.matmul(v)
        finally:
            pd.FORCE_COLLECTIVES = False
        np.save(os.path.join(outdir, "diff.npy"), np.array(float((got - want).abs().max())))
    finally:
        dist.destroy_process_group()


def test_forced_collectives_with_one_rank(tmp_path):
    mp.spawn(_forced_world1_worker, args=(1, _free_port(), 600, 3, 2, str(tmp_path)), nprocs=1, join=True)
    assert float(np.load(tmp_path / "diff.npy")) <= 1e-6


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _gather_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simplex_gp_amd.distributed import all_gather_rows
        rows = 3 + 2 * rank                                   # uneven blocks, like per-rank vertex key sets
        t = torch.full((rows, 4), rank, dtype=torch.int32) + torch.arange(rows, dtype=torch.int32)[:, None] * 10
        full, counts = all_gather_rows(t)
        assert counts == [3 + 2 * r for r in range(world)]
        want = torch.cat([torch.full((3 + 2 * r, 4), r, dtype=torch.int32)
                          + torch.arange(3 + 2 * r, dtype=torch.int32)[:, None] * 10 for r in range(world)])
        assert torch.equal(full, want)
        if rank == 0:
            np.save(os.path.join(outdir, "ok.npy"), np.array(1))
    finally:
        dist.destroy_process_group()


def test_all_gather_rows_uneven(tmp_path):
    """The key exchange of the sharded lattice build: rank blocks of different length, rank order kept."""
    mp.spawn(_gather_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert (tmp_path / "ok.npy").exists()


# ---- column-sharded batched CG and the points x columns grid (distributed.column_sharded_solve, SolveGrid) ---------

def _oracle_khat(x, s, noise):
    from oracle import oracle

    def mm(V):
        oracle.set_exact_mode(False)
        try:
            kv = oracle.filter(np.ascontiguousarray(V.numpy()), x.numpy(), RBF1)
        finally:
            oracle.set_exact_mode(True)
        return s * torch.from_numpy(kv) + noise * V
    return mm


def _column_worker(rank, world, port, n, d, t, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from simplex_gp_amd.distributed import column_sharded_solve, column_bounds
    from simplex_gp_amd.solvers import batched_cg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(21)
        x = torch.randn(n, d, generator=g)
        rhs = torch.randn(n, t, generator=g)
        mm = _oracle_khat(x, 0.7, 0.3)                      # every rank holds the whole operator (replicated build)
        solve = lambda B: batched_cg(mm, B, max_iter=40, tol=1e-5, want_tridiag=True, check_every=1)     # noqa: E731
        X, info = column_sharded_solve(solve, rhs)
        lo, hi = column_bounds(t, world, rank)
        assert info["columns"] == (lo, hi) and X.shape == (n, t)
        assert info["residual"].shape == (t,) and info["tridiag"].shape[0] == t
        Xl, il = column_sharded_solve(solve, rhs, gather=False)
        assert Xl.shape == (n, hi - lo) and torch.equal(Xl, X[:, lo:hi]) and il["exchange_bytes"] == 0
        if rank == 0:
            np.save(os.path.join(outdir, "X.npy"), X.numpy())
            np.save(os.path.join(outdir, "res.npy"), info["residual"].numpy())
            np.save(os.path.join(outdir, "T.npy"), info["tridiag"].numpy())
            np.save(os.path.join(outdir, "it.npy"), np.array(info["iterations"]))
            np.save(os.path.join(outdir, "bytes.npy"), np.array(info["exchange_bytes"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,t", [(2, 5), (3, 11), (3, 2)])
def test_column_sharded_solve_equals_single_process(tmp_path, world, t):
    """Columns of a batched CG solve split over the ranks (every rank holds the whole lattice): no collective inside the
    iteration, one all-gather at the end.  Every column block of the gathered solution, its residuals and its Lanczos
    tridiagonals are BIT-IDENTICAL to that block solved alone in one process (what a rank does; the plumbing adds
    nothing), also when a rank has no column at all (3 ranks, 2 columns); against ONE batched solve of all t columns the
    columns agree to the rounding of the dot products (torch picks the reduction order of `(a * b).sum(0)` by the
    tensor's width, and the HIP kernels are picked by vd: bitwise identity across column splits is not a property of
    either path -- independence of the columns is)."""
    from simplex_gp_amd.solvers import batched_cg
    from simplex_gp_amd.distributed import column_bounds
    n, d = 400, 2
    mp.spawn(_column_worker, args=(world, _free_port(), n, d, t, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, d, generator=g)
    rhs = torch.randn(n, t, generator=g)
    mm = _oracle_khat(x, 0.7, 0.3)
    got, res, T = np.load(tmp_path / "X.npy"), np.load(tmp_path / "res.npy"), np.load(tmp_path / "T.npy")
    k = int(np.load(tmp_path / "it.npy"))
    its = []
    for r in range(world):
        lo, hi = column_bounds(t, world, r)
        if hi == lo:
            continue
        want, info = batched_cg(mm, rhs[:, lo:hi].contiguous(), max_iter=40, tol=1e-5, want_tridiag=True, check_every=1)
        its.append(info["iterations"])
        assert np.array_equal(got[:, lo:hi], want.numpy())
        assert np.array_equal(res[lo:hi], info["residual"].numpy())
        kk = info["iterations"]
        assert np.array_equal(T[lo:hi, :kk, :kk], info["tridiag"].numpy())
        if kk < k:                                            # padded like a converged column: identity
            assert np.array_equal(T[lo:hi, kk:, kk:], np.broadcast_to(np.eye(k - kk), (hi - lo, k - kk, k - kk)))
    assert k == max(its)
    assert int(np.load(tmp_path / "bytes.npy")) == n * t * 4
    whole, info = batched_cg(mm, rhs, max_iter=40, tol=1e-5, want_tridiag=True, check_every=1)
    assert np.linalg.norm(got - whole.numpy()) / np.linalg.norm(whole.numpy()) <= 1e-5
    assert float(res.max()) <= 1e-3                           # 40 iterations at most


def _grid_worker(rank, world, port, n, d, t, C, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.oracle_lattice_adapter import OracleLattice
    from simplex_gp_amd.distributed import SolveGrid
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(22)
        x = torch.randn(n, d, generator=g)
        rhs = torch.randn(n, t, generator=g)
        grid = SolveGrid(C)
        assert (grid.C, grid.P) == (C, world // C) and grid.rank == grid.c * grid.P + grid.p
        lo, hi = grid.rows(n)
        op = ShardedLatticeMVM.from_local_rows(x[lo:hi].contiguous(), RBF1, group=grid.point_group, lattice=OracleLattice(),
                                               n_total=n)
        assert (op.lo, op.hi) == (lo, hi)
        X_rows, info = grid.solve(op, rhs[lo:hi].contiguous(), 0.7, 0.3, max_iter=200, tol=1e-6)
        assert X_rows.shape == (hi - lo, t)
        np.save(os.path.join(outdir, f"X{rank}.npy"), X_rows.numpy())
        np.save(os.path.join(outdir, f"rows{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,C", [(4, 2), (3, 3), (2, 1)])
def test_solve_grid_points_by_columns(tmp_path, world, C):
    """C column groups x P point shards (4 ranks as 2 x 2; 3 ranks as pure column sharding; 2 ranks as pure point
    sharding): every rank ends up with its rows of ALL columns, and the assembled solution satisfies the single-process
    operator equation."""
    from oracle import oracle
    n, d, t = 601, 3, 5
    mp.spawn(_grid_worker, args=(world, _free_port(), n, d, t, C, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(22)
    x = torch.randn(n, d, generator=g)
    rhs = torch.randn(n, t, generator=g)
    P = world // C
    sol = np.zeros((n, t), np.float32)
    for rank in range(world):
        lo, hi = np.load(tmp_path / f"rows{rank}.npy")
        block = np.load(tmp_path / f"X{rank}.npy")
        if rank < P:
            sol[lo:hi] = block
        else:                                   # the other column groups hold the same rows: identical copies
            assert np.array_equal(sol[lo:hi], block)
    oracle.set_exact_mode(False)
    try:
        back = 0.7 * oracle.filter(sol, x.numpy(), RBF1) + 0.3 * sol
    finally:
        oracle.set_exact_mode(True)
    assert np.linalg.norm(back - rhs.numpy()) / np.linalg.norm(rhs.numpy()) <= 1e-4


def _mll_worker(rank, world, port, n, pre_size, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    from simplex_gp_amd.distributed import column_sharded_mll, all_reduce_gradients
    from oracle import oracle

    def oracle_filter(src, ref, coeffs):
        return torch.from_numpy(oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy()))
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        x = torch.rand(n, 2) * 3
        y = torch.sin(2 * x[:, 0]) * torch.cos(x[:, 1]) + 0.1 * torch.randn(n)
        model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=2), min_noise=1e-2)
        out = column_sharded_mll(model, x, y, num_probes=7, cg_tol=1e-6, max_cg_iter=300, seed=5, pre_size=pre_size)
        (-out).backward()
        all_reduce_gradients(model)
        if rank == 0:
            np.save(os.path.join(outdir, "value.npy"), np.array(float(out.detach())))
            np.save(os.path.join(outdir, "grads.npy"), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy())
    finally:
        dist.destroy_process_group()
        plx.LatticeFilterGeneral.method = None


@pytest.mark.parametrize("world,pre_size", [(2, 0), (3, 0), (2, 10)])
def test_column_sharded_training_step_equals_single_process(tmp_path, world, pre_size):
    """One marginal-likelihood evaluation + backward with the [y | probes] columns sharded over the ranks (replicated lattice,
    same probes from the same seed, one all-reduce of two scalars for the value and one of the hyper-parameter gradients):
    the value and every gradient equal the single-process solvers.marginal_log_likelihood."""
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    from oracle import oracle
    n = 300
    mp.spawn(_mll_worker, args=(world, _free_port(), n, pre_size, str(tmp_path)), nprocs=world, join=True)

    def oracle_filter(src, ref, coeffs):
        return torch.from_numpy(oracle.filter(src.detach().numpy(), ref.detach().numpy(), coeffs.detach().numpy()))
    plx.LatticeFilterGeneral.method = staticmethod(oracle_filter)
    try:
        torch.manual_seed(3)
        x = torch.rand(n, 2) * 3
        y = torch.sin(2 * x[:, 0]) * torch.cos(x[:, 1]) + 0.1 * torch.randn(n)
        model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=2), min_noise=1e-2)
        want = solvers.marginal_log_likelihood(model, x, y, num_probes=7, cg_tol=1e-6, max_cg_iter=300, seed=5, pre_size=pre_size)
        (-want).backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    finally:
        plx.LatticeFilterGeneral.method = None
    assert abs(float(np.load(tmp_path / "value.npy")) - float(want.detach())) <= 1e-5 * (1 + abs(float(want.detach())))
    got = np.load(tmp_path / "grads.npy")
    assert np.allclose(got, grads, rtol=2e-4, atol=1e-6), (got, grads)


def test_bench_expected_speedup_model():
    """bench.expected_speedup: the prediction a multi-rank bench line carries next to its measurement (DESIGN.md 5's model:
    splat and slice shard by points, the blur is replicated, the exchange exists only between ranks)."""
    import bench
    st = {"splat": 40.0, "exchange": 50.0, "blur": 44.0, "slice": 30.0}           # the config-4 figures of DESIGN.md 5, us
    e = bench.expected_speedup(st, 8)
    assert e["t_rank_us"] == 164.0 and e["t_one_gpu_us"] == 8 * 70.0 + 44.0
    assert abs(e["warm_mvm"] - 604.0 / 164.0) < 0.01 and abs(e["warm_mvm_if_exchange_were_free"] - 604.0 / 114.0) < 0.01
    assert abs(e["replicated_share_of_rank_time"] - 94.0 / 164.0) < 1e-3
    # no replicated work and a free exchange: the model's ceiling is the rank count
    assert bench.expected_speedup({"splat": 10.0, "exchange": 0.0, "blur": 0.0, "slice": 10.0}, 8)["warm_mvm"] == 8.0
    # the cadence with a build that does not shard: one build of 1 ms per 20 MVMs
    b = bench.expected_speedup(st, 8, build=(1.0, 20))
    assert abs(b["with_one_build_per_20_mvms"] - (604.0 + 50.0) / (164.0 + 50.0)) < 0.01
    assert bench.expected_speedup({}, 8) is None
