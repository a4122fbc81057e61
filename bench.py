#!/usr/bin/env python3
"""bench.py -- lattice K.v MVMs/sec on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2] -- synthetic N=1e6, d=8,
RBFLattice order=1 (taps [0.34608543, 1, 0.34608543]), vd=1, lengthscale 1.0,
x ~ N(0, I) from torch.Generator().manual_seed(1234), run the way
the reference's CG loop drives it: ONE lattice build per `--rebuild-every`
(default 50) MVMs, i.e. the timed region of a default run is 1 build + 50
applies.  A "step" is one K.v MVM through the C ABI (simplex_gp_amd ->
libplx.so); step i rebuilds the lattice first when i % rebuild_every == 0, so
no work is skipped: `value` = steps / wall time, inputs resident in HBM.

Also reported on the same JSON line (extra keys):
  cold_mvms_per_s  every step = filter(src, ref, coeffs) = build + apply, what
                   the reference does on every call (permutohedral.h:272)
  warm_mvms_per_s  apply only
  roofline         dominant kernel of the timed region: algorithmic bytes per
                   launch (SURVEY 8d formulas) / mean launch time from hipEvents
                   recorded by plx_apply on its own stream
  fine             the same lattice shape at lengthscale 0.25 (m ~ 8.9e6), where
                   the blur stage streams from HBM: blur roofline fraction
  cpu_baseline     the reference's own CPU extension (oracle/_ref, built from
                   /root/reference in the dev container) or, if absent, the C
                   port (oracle/), timed on one full-size MVM on this host

Multi-GPU (launched by torch.distributed.run, one rank per GPU): weak scaling,
n = 1e6 points PER GPU.  Build: every rank embeds / inserts its own rows, one
all-gather of the per-rank vertex keys, merge into one numbering.  MVM: splat own
rows, one RCCL all-reduce of the vertex accumulators, replicated blur, slice own
rows.  value = (n_total / 1e6) * MVMs/s, i.e. 1e6-point
row blocks of K.v produced per second by the whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RBF1 = np.array([0.34608543, 1.0, 0.34608543], np.float32)
HBM_PEAK_GBPS = 8000.0       # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def alg_bytes(n, d, m, vd, r):
    """compulsory bytes per launch, fp32 values / int32 ids."""
    return {
        "splat": 4 * n * vd + 8 * n * (d + 1) + 4 * m * vd,
        "blur_axis": m * (8 * vd + 8 * r),
        "slice": 8 * n * (d + 1) + 4 * m * vd + 4 * n * vd,
    }


def synth(n, d, vd, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g)
    v = torch.randn(n, vd, generator=g)
    return x, v


def time_region(fn, steps, sync, barrier):
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    sync()
    barrier()
    return time.perf_counter() - t0


def kernel_times(lat, v, out, reps):
    """Mean device time (ms) per LAUNCH of each apply kernel.  plx_apply records one
    hipEvent pair per stage on its own stream; the blur stage is d+1 back-to-back
    launches of one kernel, so stage / (d+1) is that kernel's mean launch time
    (agrees with rocprofv3's per-kernel average to a few percent; per-launch
    event pairs would add ~20 % of event overhead to a 25 us kernel)."""
    lat.set_timing(True)
    acc = {"splat": [], "blur": [], "slice": []}
    for _ in range(reps):
        lat.apply(v, out)
        t = lat.apply_times_ms()
        for k in acc:
            acc[k].append(t[k])
    lat.set_timing(False)
    return {"splat": float(np.mean(acc["splat"])), "blur": float(np.mean(acc["blur"])) / (lat.d + 1),
            "slice": float(np.mean(acc["slice"]))}


def roofline_for(kt, n, d, m, vd, r, ell=1.0):
    ab = alg_bytes(n, d, m, vd, r)
    per_mvm_ms = {"splat": kt["splat"], "blur_axis": kt["blur"] * (d + 1), "slice": kt["slice"]}
    dom = max(per_mvm_ms, key=per_mvm_ms.get)
    launch_ms = {"splat": kt["splat"], "blur_axis": kt["blur"], "slice": kt["slice"]}[dom]
    achieved = ab[dom] / (launch_ms * 1e-3) / 1e9
    kernel = {"splat": "splat_scan_kernel (+gather_in, fix-up)", "blur_axis": "blur_axis kernel", "slice": "slice_v1_kernel"}[dom]
    prefix = {"splat": "plx::splat_scan_kernel", "blur_axis": "plx::blur_axis", "slice": "plx::slice_v1_kernel"}[dom]
    grid = {"splat": 8 * ((-(-n * (d + 1) // 1024) + 7) // 8) * 256,
            "blur_axis": 8 * ((-(-(-(-m // 4)) // 256) + 7) // 8) * 256,
            "slice": 8 * ((-(-n // 256) + 7) // 8) * 256}[dom]
    pmc = pmc_traffic(prefix, grid, ell)
    return {
        "bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": pmc["bytes"] if pmc else None, "traffic_source": pmc["source"] if pmc else None,
        "bytes_per_launch": int(ab[dom]), "launch_us": round(launch_ms * 1e3, 2),
        "launches_per_mvm": (d + 1) if dom == "blur_axis" else 1,
        "note": ("cache-resident lattice: bound by the L2 request rate of 4-byte gathers, not HBM (DESIGN.md 4); "
                 "the HBM-bound regime is reported under 'fine'") if m * (d + 1) * 8 * r < 128e6 else "",
    }, {
        k: {"us_per_mvm": round(per_mvm_ms[k] * 1e3, 2), "alg_MB_per_mvm": round(ab[k] * ((d + 1) if k == "blur_axis" else 1) / 1e6, 2),
            "GBps": round(ab[k] * ((d + 1) if k == "blur_axis" else 1) / (per_mvm_ms[k] * 1e-3) / 1e9, 1)}
        for k in per_mvm_ms
    }


def pmc_traffic(kernel_prefix, grid, ell):
    """HBM bytes per launch of a kernel from the newest committed PMC table (profiles/*_pmc.json, written by
    tools/summarize_profile.py from separate rocprofv3 --pmc passes), corrected as the microarch guide
    prescribes for gfx950 (FETCH_SIZE x2 + WRITE_SIZE).  None when no table has that kernel at that grid."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            table = json.load(open(path))["by_lengthscale"].get(str(ell), {})
        except Exception:                      # noqa: BLE001
            continue
        for key, rec in table.items():
            name, g = key.rsplit("|", 1)
            if name.startswith(kernel_prefix) and int(g) == grid:
                best = {"bytes": int((2 * rec["fetch_KB"] + rec["write_KB"]) * 1024), "source": os.path.basename(path)}
    return best


def cpu_baseline(x, v, ell):
    """One full-size MVM on the host: the reference's own extension if oracle/_ref
    is present (kind "reference"), else the C port (kind "port")."""
    ref = (x / ell).contiguous()
    taps = torch.from_numpy(RBF1)
    try:
        from oracle import build_ref
        mod = build_ref.load("cpu_lattice_ref")
        kind, fn = "reference", (lambda: mod.filter(v, ref, taps))
    except Exception as e:                      # noqa: BLE001
        log("cpu_baseline: oracle/_ref unavailable (%s); timing the C port" % e)
        from oracle import oracle
        kind, fn = "port", (lambda: oracle.filter(v.numpy(), ref.numpy(), RBF1))
    torch.set_num_threads(1)
    best = float("inf")
    t_start = time.perf_counter()
    reps = 0
    while reps < 3 and time.perf_counter() - t_start < 25:
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return {"value": round(1.0 / best, 4), "unit": "MVMs/s", "cores": 1, "kind": kind,
            "host_cores": os.cpu_count(),
            "sample": f"{reps} full filter() call(s) at N={x.shape[0]}, d={x.shape[1]}, vd={v.shape[1]}, "
                      f"lengthscale {ell}; best taken; single thread (the reference CPU path is single-threaded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", dest="n", type=int, default=1_000_000, help="points per GPU")
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--vd", type=int, default=1)
    ap.add_argument("--ell", type=float, default=1.0)
    ap.add_argument("--rebuild-every", type=int, default=50)
    ap.add_argument("--skip-cpu-baseline", dest="no_cpu_baseline", action="store_true")
    ap.add_argument("--skip-fine", dest="no_fine", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo to rehearse ranks on one GPU")
    ap.add_argument("--dump", default=None, help="write every rank's output rows of one MVM to <DUMP>.rank<r>.npz (checked by tests/check_bench_dump.py)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import simplex_gp_amd as plx
    from simplex_gp_amd import _native as nv

    def sync():
        torch.cuda.synchronize(dev)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[dev_index]) if args.backend == "nccl" else dist.barrier()

    n_local, d, vd, r = args.n, args.d, args.vd, 1
    n_total = n_local * world
    x, v_all = synth(n_total, d, vd)
    lo, hi = rank * n_local, (rank + 1) * n_local
    # a rank keeps only its own rows on the device (the sharded build never needs the others)
    ref = (x[lo:hi] / args.ell).contiguous().to(dev) if world > 1 else (x / args.ell).contiguous().to(dev)
    v = v_all[lo:hi].contiguous().to(dev)
    out = torch.empty_like(v)

    lat = plx.Lattice(dev)

    def build():
        """One lattice build.  Multi-GPU: every rank embeds / inserts only its own rows, the per-rank
        vertex keys are all-gathered (the one collective of a build) and merged into one numbering."""
        if world == 1:
            lat.build(ref, RBF1)
        else:
            keys = lat.build_local(ref_local, RBF1)
            all_keys, counts = all_gather_rows(keys)
            lat.build_merge(all_keys, counts, rank)

    if world > 1:
        from simplex_gp_amd.distributed import all_gather_rows
        ref_local = ref
    build()
    m = lat.m
    values = scratch = None
    if world > 1:
        values, scratch = lat.new_values(vd), lat.new_values(vd)

    def mvm():
        if world == 1:
            lat.apply(v, out)
        else:
            lat.splat(v, values)
            dist.all_reduce(values)            # RCCL sum over xGMI: the one exchange of the path
            res = lat.blur(values, scratch, vd=vd)
            lat.slice(res, out, vd=vd)

    def step(i):
        if i % args.rebuild_every == 0:
            build()
        mvm()

    if args.dump:
        # rows of one MVM for an external checker (tests/check_bench_dump.py compares them with the CPU oracle);
        # bench.py itself only touches oracle/ in the cpu_baseline leg
        mvm()
        sync()
        np.savez(f"{args.dump}.rank{rank}.npz", out=out.cpu().numpy(), lo=lo, hi=hi, n_total=n_total, d=d, vd=vd,
                 ell=args.ell, m=m)

    for i in range(args.warmup):
        step(i)
    wall = time_region(step, args.steps, sync, barrier)
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    mvms_per_s = args.steps / wall
    value = mvms_per_s * (n_total / 1e6) if world > 1 else mvms_per_s

    result = {
        "metric": "lattice K.v MVMs/sec, N=1e6 d=8 order=1 (1e6 points per GPU)",
        "value": round(value, 2), "unit": "MVMs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"N={n_local} points/GPU x {world} GPU(s), d={d}, vd={vd}, RBFLattice order=1, "
                        f"lengthscale {args.ell}, x~N(0,I) seed 1234; one lattice build per "
                        f"{args.rebuild_every} MVMs (CG loop, BASELINE.json configs[2])",
            "n_total": n_total, "m_vertices": m, "rebuild_every": args.rebuild_every,
            "parallelism": "single GPU" if world == 1 else f"points sharded x{world}; build: local + all-gather of vertex "
                                                              "keys + merge; MVM: RCCL all-reduce of vertex values, replicated blur",
            "value_definition": "MVMs/s" if world == 1 else "(n_total/1e6) x MVMs/s of the n_total-point operator",
        },
    }

    if world == 1:
        # ---- warm / cold rates and per-kernel times on the same lattice
        wall_warm = time_region(lambda i: mvm(), args.steps, sync, barrier)
        wall_cold = time_region(lambda i: (lat.build(ref, RBF1), mvm()), max(5, args.steps // 5), sync, barrier)
        lat.set_timing(True)
        lat.build(ref, RBF1)
        build_ms = lat.build_times_ms()
        lat.set_timing(False)
        kt = kernel_times(lat, v, out, reps=max(10, args.steps))
        roof, stages = roofline_for(kt, n_local, d, m, vd, r, args.ell)
        result["warm_mvms_per_s"] = round(args.steps / wall_warm, 1)
        result["cold_mvms_per_s"] = round(max(5, args.steps // 5) / wall_cold, 1)
        result["roofline"] = roof
        result["stages"] = stages
        result["build_ms"] = {k: round(t, 3) for k, t in build_ms.items()}
        result["lattice_device_MB"] = round(lat.device_bytes / 1e6, 1)
        log("kernel times (ms):", kt)
        log("build (ms):", build_ms)

        if not args.no_fine:
            # ---- fine regime: same points, lengthscale 0.25 -> m ~ 8.9e6, blur streams from HBM
            ref_f = (x / 0.25).contiguous().to(dev)
            lat_f = plx.Lattice(dev)
            lat_f.set_timing(True)
            lat_f.build(ref_f, RBF1)
            fine_build = lat_f.build_times_ms()
            lat_f.set_timing(False)
            for _ in range(3):
                lat_f.apply(v, out)
            wf = time_region(lambda i: lat_f.apply(v, out), 20, sync, barrier)
            ktf = kernel_times(lat_f, v, out, reps=10)
            abf = alg_bytes(n_local, d, lat_f.m, vd, r)
            blur_gbps = abf["blur_axis"] / (ktf["blur"] * 1e-3) / 1e9
            _, stages_f = roofline_for(ktf, n_local, d, lat_f.m, vd, r)
            result["fine"] = {
                "lengthscale": 0.25, "m_vertices": lat_f.m, "warm_mvms_per_s": round(20 / wf, 1),
                "blur_roofline": {"bound": "hbm", "kernel": "blur_axis_kernel", "achieved": round(blur_gbps, 1),
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(blur_gbps / HBM_PEAK_GBPS, 4),
                                  "bytes_per_launch": int(abf["blur_axis"]), "launch_us": round(ktf["blur"] * 1e3, 2)},
                "stages": stages_f, "build_ms": {k: round(t, 3) for k, t in fine_build.items()},
                "lattice_device_MB": round(lat_f.device_bytes / 1e6, 1),
            }
            log("fine kernel times (ms):", ktf)
            lat_f.close()
            del ref_f

        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(x[:n_local], v_all[:n_local], args.ell)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
