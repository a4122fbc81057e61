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
