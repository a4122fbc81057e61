#!/usr/bin/env python3
"""Round 5: where the pivoted-Cholesky factor's wall time goes (config-3 size: N=1e6, d=8, l=0.6931, rank 100).
The constructor's batch loop, piece by piece: (a) with a synchronisation after every piece = device time per piece,
(b) without = host enqueue time per piece, next to the constructor's own wall time.

    python tools/pchol_factor_timing.py [n] [rank]
"""
import ctypes, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from simplex_gp_amd import solvers, _native as nv

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g).cuda()
model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).cuda()
lib = nv.lib()
_vp = solvers._vp


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def loop(lat, s, k, do_sync, sparse=True, batch=12, rel_tol=1e-6):
    dev = lat.device
    nn = lat.n_owned
    kp = max(16, (k + 15) // 16 * 16)
    ld = (nn + 63) // 64 * 64
    acc = {}
    def tick(name, t0):
        t1 = sync() if do_sync else time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1
    t0 = sync()
    Lt = torch.zeros(kp, ld, dtype=torch.float32, device=dev)
    diag = torch.full((nn,), s, dtype=torch.float32, device=dev)
    row_rank = torch.empty(lat.n, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nv.check(lib.plx_copy_point_perm(lat._h, _vp(row_rank), stream), "perm")
    work = torch.empty(int(lib.plx_pchol_work_bytes(ld, kp)), dtype=torch.uint8, device=dev)
    cand = torch.empty(16, dtype=torch.int32, device=dev)
    accepted = torch.zeros(2, dtype=torch.int32, device=dev)
    exact = False
    frontier = torch.zeros(1, dtype=torch.int32, device=dev)
    fronts = []
    scale = torch.tensor([s, 1.0], dtype=torch.float32, device=dev)
    was = lat.lattice_rows
    lat.set_lattice_row_order(True)
    bufs = {}
    m, B, batches = 0, batch, 0
    t = tick("setup", t0)
    while m < k:
        nb = min(B, k - m, nn)
        tt = nb if nb == 1 else (nb + 3) // 4 * 4
        if tt not in bufs:
            bufs[tt] = (torch.empty(nn, tt, dtype=torch.float32, device=dev), lat.new_values(tt), lat.new_values(tt))
        rows, vals, scratch = bufs[tt]
        t = tick("bufs", t)
        nv.check(lib.plx_pchol_select(_vp(diag), _vp(row_rank), nn, nb, ld, kp, _vp(cand), _vp(work), stream), "select")
        t = tick("select", t)
        if sparse is None:
            lat.splat_onehot(cand, nb, vals, vd=tt)
            t = tick("splat_onehot", t)
            bl = lat.blur(vals, scratch, vd=tt)
            t = tick("blur", t)
            lat.slice(bl, rows, vd=tt)
            t = tick("slice", t)
        else:
            lat.filter_onehot(cand, nb, vals, scratch, rows, vd=tt, sparse=sparse, frontier=frontier)
            t = tick("filter_onehot", t)
        nv.check(lib.plx_pchol_factor_batch(_vp(Lt), ld, kp, m, _vp(rows), tt, _vp(scale), _vp(cand), nb, _vp(diag), _vp(row_rank), nn,
                                            float(rel_tol * s), int(exact), _vp(accepted), _vp(work), stream), "factor_batch")
        t = tick("factor_batch", t)
        a, planned = accepted.tolist()
        exact = planned < nb
        fronts.append(int(frontier.item()))
        t = tick("readback", t)
        m += a
        batches += 1
        B = min(batch, 16, 2 * a) if a < nb else min(batch, 16, max(B, 2 * a))
    lat.set_lattice_row_order(was)
    t1 = sync()
    return {"sync_each_piece": do_sync, "kernel_rows": {None: "dense stages", True: "frontier", False: "dense (one call)"}[sparse], "batches": batches, "m": lat.m, "frontier_max": max(fronts), "wall_ms": round((t1 - t0) * 1e3, 3),
            "pieces_ms": {k_: round(v * 1e3, 3) for k_, v in acc.items()}}


with torch.no_grad():
    K = model.kernel(x, x)
    lat0 = model.preconditioner(x, rank, K=K).lat
    for sp in (True, False, True, False):
        for rep in range(3):
            t0 = sync()
            pre = solvers.LatticePreconditioner(lat0, float(model.outputscale), float(model.noise), rank, sparse_rows=sp)
            t1 = sync()
            print(json.dumps({"constructor_ms": round((t1 - t0) * 1e3, 2), "sparse_rows": sp, "batches": pre.batches,
                              "sparse_batches": pre.sparse_batches}), flush=True)
    lat = pre.lat
    s = pre.outputscale
    for do_sync in (True, False):
        for sp in (None, True):
            print(json.dumps(loop(lat, s, rank, do_sync, sparse=sp)), flush=True)
