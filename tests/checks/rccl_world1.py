#!/usr/bin/env python3
"""One rank, one GPU, backend nccl (= RCCL): the calls an 8-GPU job makes, each executed once on real hardware.

    init_process_group("nccl", device_id=...)            bench.py Ctx
    all_gather of per-rank meta + padded vertex keys      distributed.all_gather_rows (sharded build)
    all_reduce(sum) of the vertex accumulator [m, vdp]    ShardedLatticeMVM.matmul
    all_reduce of CG dot products                         distributed.sharded_solve
    barrier(device_ids=[...])                             bench.py Ctx.barrier

and the sharded operator built from local rows must equal the plain single-GPU lattice BIT FOR BIT (one rank: the
merged numbering is the plain one up to a relabelling, which no summation order depends on).  Run as a child process by
tests/test_baseline_configs.py::test_rccl_world_size_one; prints RCCL_WORLD1_OK."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

import simplex_gp_amd as plx
from simplex_gp_amd import distributed as pd

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
pd.FORCE_COLLECTIVES = True

taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
g = torch.Generator().manual_seed(1234)
n, d = 200_000, 8
x = torch.randn(n, d, generator=g).to(dev)
ok = True
for vd in (1, 11):
    v = torch.randn(n, vd, generator=g).to(dev)
    plain = plx.Lattice(dev).build(x, taps)
    want = plain.apply(v)
    op = pd.ShardedLatticeMVM.from_local_rows(x, taps, n_total=n)
    assert op.world == 1 and op.key_bytes_exchanged > 0, "the key all-gather must have run"
    assert op.m == plain.m
    got = op.matmul(v)
    dist.barrier(device_ids=[0])
    same = torch.equal(got, want)
    print(f"vd={vd}: m={op.m} exchange {op.exchange_bytes(vd)} bytes, bit-identical to the plain lattice: {same}", flush=True)
    ok = ok and same
    plain.close()
# a CG solve whose dot products go through all_reduce
v = torch.randn(n, 3, generator=g).to(dev)
op = pd.ShardedLatticeMVM.from_local_rows(x, taps, n_total=n)
sol, info = pd.sharded_solve(op, v, 0.7, 0.5, max_iter=30, tol=1e-4)
resid = v - (0.7 * op.matmul(sol) + 0.5 * sol)
rel = float((resid.norm(dim=0) / v.norm(dim=0)).max())
# the same solve with the collectives off: one rank, so the all-reduced dot products must not change a bit
pd.FORCE_COLLECTIVES = False
sol_local, _ = pd.sharded_solve(op, v, 0.7, 0.5, max_iter=30, tol=1e-4)
pd.FORCE_COLLECTIVES = True
same = torch.equal(sol, sol_local)
print(f"sharded CG over RCCL: {info['iterations']} iterations, true relative residual {rel:.2e} (the lattice operator is only "
      f"approximately symmetric: CG stalls near 1e-1 .. 1e-2), equal to the solve without collectives: {same}", flush=True)
ok = ok and same and rel < 0.5
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WORLD1_OK" if ok else "RCCL_WORLD1_MISMATCH", flush=True)
sys.exit(0 if ok else 1)
