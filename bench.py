#!/usr/bin/env python3
"""bench.py -- lattice K.v MVMs/sec on MI355X (BASELINE.json metric).

Headline workload (config.workload): synthetic N=1e6 points, d=8,
RBFLattice order=1 (taps [0.34608543, 1, 0.34608543]), vd=1, lengthscale 1.0,
x ~ N(0, I) from torch.Generator().manual_seed(1234), driven the way
the reference's CG loop drives it (BASELINE.json configs[2]): ONE lattice build
per `--rebuild-every` (default 50) MVMs.  A "step" is one K.v MVM through the C
ABI (simplex_gp_amd -> libplx.so); step i rebuilds the lattice first when
i % rebuild_every == 0.  `value` = steps / wall time of the timed region, inputs
resident in HBM; the label reports how many builds the timed region contained.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong|config4]

With --gpus N > 1 and no launcher in the environment, bench.py starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
process (before anything touches the GPU) and relays its JSON line and exit code.

Extra keys on the same JSON line (one GPU):
  warm_mvms_per_s / cold_mvms_per_s   apply only / build + apply per call (what the
                   reference does on every call, permutohedral.h:272)
  roofline         dominant stage of the timed region: algorithmic bytes
                   / mean stage time from hipEvents recorded by plx_apply on its stream
  stages           the same for all three stages
  fine             the same shape at lengthscale 0.25 (m ~ 8.9e6): the blur streams
                   from HBM there; its roofline fraction is the north-star target
  config3_cg_ms    BASELINE.json configs[2]: N=1e6, d=8, lengthscale 0.6931, 50 CG
                   iterations on (sK + sigma^2 I) with [y | 10 probes] incl. the build;
                   config3.factor: the rank-100 pivoted-Cholesky factor on that lattice;
                   config3.train_step: one Adam step of the reference's recipe
  config4          N=4e6 total, d=8, lengthscale 1 (configs[3]) sharded over the ranks
  config5_mvm_us   MaternLattice(nu=1.5, order=3), N=10,623, d=18 stand-in: one MVM
  cpu_baseline     the reference's own CPU extension (oracle/_ref) or the C port,
                   one full-size MVM on this host, single thread

Multi-GPU (one rank per GPU, RCCL): points are sharded by contiguous row blocks.
Build: every rank embeds / inserts its own rows, ONE all-gather of the per-rank
vertex keys, merge into one numbering.  MVM: splat own rows, ONE all-reduce of the
vertex accumulators, replicated blur, slice own rows.  `value` is ALWAYS the plain
rate of one operator, MVMs/s -- never scaled by the problem size:
  --scaling strong  (default) N=1e6 TOTAL, BASELINE.json's metric as written
  --scaling config4 N=4e6 TOTAL (BASELINE.json configs[3])
  --scaling weak    1e6 points PER GPU: the operator grows with the rank count
Whatever the mode, a multi-GPU line also carries the other legs (`strong`, `config4`
with vd 1 and 11, `weak_1e6_per_gpu`, `weak_4e6_per_gpu`), each with per-stage us per
rank and `exchange: {kind, bytes, us}`, so that one 1/2/4/8 sweep yields every curve;
`columns_mode` (every rank applies the whole operator to its own right-hand-side column:
the batched solve's columns sharded, no collective), `config3_cg` (one batched solve split
by rows / by columns, and a column-sharded training step) and `config4.grid_vd11` (the
points x columns grid) report the modes of distributed.py that shard columns.
PLX_BENCH_SINGLE_RANK_RCCL=1 makes a one-GPU run open a world-size-1 "nccl" group and
take the multi-rank code path (RCCL calls included) -- a rehearsal switch, not a mode.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

T_START = time.perf_counter()      # (multi-rank runs report wall seconds per leg: leg_wall_s)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RBF1 = np.array([0.34608543, 1.0, 0.34608543], np.float32)
HBM_PEAK_GBPS = 8000.0       # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", dest="n", type=int, default=1_000_000, help="points in total (strong) / per GPU (weak)")
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--vd", type=int, default=1)
    ap.add_argument("--ell", type=float, default=1.0)
    ap.add_argument("--rebuild-every", type=int, default=50)
    ap.add_argument("--scaling", choices=["weak", "strong", "config4"], default="strong")
    ap.add_argument("--skip-cpu-baseline", dest="no_cpu_baseline", action="store_true")
    ap.add_argument("--skip-fine", dest="no_fine", action="store_true")
    ap.add_argument("--skip-configs", dest="no_configs", action="store_true", help="skip the config 3/4/5 legs")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo to rehearse ranks on one GPU")
    ap.add_argument("--skip-weak4", dest="no_weak4", action="store_true", help="multi-rank runs: skip the 4e6-points-per-GPU weak-scaling leg")
    ap.add_argument("--dump", default=None, help="write every rank's output rows of one MVM to <DUMP>.rank<r>.npz (checked by tests/check_bench_dump.py)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the ranks as a child torchrun and relay its exit code.
    Nothing in this process has touched the GPU (torch is not even imported yet), so no exec is involved."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: no launcher in the environment; starting", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def alg_bytes(n, d, m, vd, r):
    """compulsory bytes per launch, fp32 values / int32 ids."""
    return {
        "splat": 4 * n * vd + 8 * n * (d + 1) + 4 * m * vd,
        "blur_axis": m * (8 * vd + 8 * r),
        "slice": 8 * n * (d + 1) + 4 * m * vd + 4 * n * vd,
    }


SYNTH_BLOCK = 1_000_000
PREWARM_MVMS = 200          # untimed MVMs before the warm-up steps (about 20 ms: brings the GPU to its steady clocks)


def synth(n, d, vd, seed=1234, lo=0, hi=None):
    """Rows [lo, hi) of the synthetic data set of n rows: x ~ N(0, I_d), v ~ N(0, 1), generated in blocks of 1e6 rows,
    block b from torch.Generator().manual_seed(seed + b) (x drawn first, then v: block 0 is SURVEY 8d's recipe).  Every
    rank of a sharded job generates only the blocks its rows live in, whatever the total size."""
    import torch
    hi = n if hi is None else hi
    xs, vs = [], []
    for b in range(lo // SYNTH_BLOCK, max(lo // SYNTH_BLOCK + 1, -(-hi // SYNTH_BLOCK))):
        b0, b1 = b * SYNTH_BLOCK, min((b + 1) * SYNTH_BLOCK, n)
        if b1 <= b0:
            break
        g = torch.Generator().manual_seed(seed + b)
        x = torch.randn(b1 - b0, d, generator=g)
        v = torch.randn(b1 - b0, vd, generator=g)
        a, z = max(lo, b0) - b0, min(hi, b1) - b0
        xs.append(x[a:z])
        vs.append(v[a:z])
    return torch.cat(xs, 0), torch.cat(vs, 0)


def time_region(fn, steps, sync, barrier):
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    sync()
    barrier()
    return time.perf_counter() - t0


def kernel_times(lat, v, out, reps, mvm_ms=None):
    """Mean device time (ms) per stage of plx_apply and per blur launch.  plx_apply records one hipEvent pair per
    stage on its own stream, inside real MVMs (so every stage sees the cache state the previous stage left, which a
    loop over one stage alone does not: the fine-regime blur runs 15 % faster in isolation); the blur stage is d+1
    back-to-back launches of one kernel family, so stage / (d+1) is the mean per-axis time.  The marker packets between
    the stages cost ~2.5 us each (three stages: +8 % on a 93 us MVM), so the stage times are scaled by
    mvm_ms / (their sum), mvm_ms = the un-instrumented warm MVM measured by the caller: the shares come from the
    events, the total from the plain loop -- this is what agrees with rocprofv3's per-kernel averages (profiles/)."""
    lat.set_timing(True)
    acc = {"splat": [], "blur": [], "slice": []}
    for _ in range(reps):
        lat.apply(v, out)
        t = lat.apply_times_ms()
        for k in acc:
            acc[k].append(t[k])
    lat.set_timing(False)
    st = {k: float(np.mean(t)) for k, t in acc.items()}
    scale = 1.0
    if mvm_ms is not None:
        scale = min(1.0, mvm_ms / sum(st.values()))
    return {"splat": st["splat"] * scale, "blur": st["blur"] * scale / (lat.d + 1), "slice": st["slice"] * scale,
            "event_overhead_scale": round(scale, 4)}


def roofline_for(lat, kt, n, d, m, vd, r, ell=1.0):
    """roofline object for the dominant stage + per-stage table.  Kernel names are the ones libplx launched for this
    lattice (plx_stage_kernels), i.e. what rocprofv3 --kernel-trace shows."""
    ab = alg_bytes(n, d, m, vd, r)
    per_mvm_ms = {"splat": kt["splat"], "blur_axis": kt["blur"] * (d + 1), "slice": kt["slice"]}
    dom = max(per_mvm_ms, key=per_mvm_ms.get)
    launch_ms = {"splat": kt["splat"], "blur_axis": kt["blur"], "slice": kt["slice"]}[dom]
    achieved = ab[dom] / (launch_ms * 1e-3) / 1e9
    names = lat.stage_kernels(vd)

    def stage_traffic(stage):
        """HBM bytes per launch-equivalent of a stage from the committed PMC table.  splat / slice: the sum over the
        stage's kernels (one launch each per MVM).  blur: per AXIS -- a two-axes-per-launch kernel counts for two."""
        ks = names[stage]
        if stage == "blur_axis" and "blur_pair_v1_kernel" in ks:
            pair, single = pmc_traffic(["blur_pair_v1_kernel"], ell), pmc_traffic(["blur_axis_v1_kernel"], ell)
            npair, nsingle = (d + 1) // 2, (d + 1) % 2
            if pair is None or (nsingle and single is None):
                return None
            total = npair * pair["bytes"] + (nsingle * single["bytes"] if nsingle else 0)
            return {"bytes": int(total / (d + 1)), "source": pair["source"]}
        return pmc_traffic(ks, ell)

    pmc = stage_traffic(dom)
    cache_resident = m * (d + 1) * 8 * r < 128e6
    roof = {
        "bound": "hbm", "stage": dom, "kernel": " + ".join(names[dom]), "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": pmc["bytes"] if pmc else None,
        # (a PMC table is only quoted for the kernel sources it was collected on: an edited source says so here instead of
        # turning the figure into a silent null)
        "traffic_source": pmc["source"] if pmc else ("none: no profiles/*_pmc.json was collected on these kernel sources (sha16 %s); "
                                                     "run tools/run_profiles.sh" % kernel_sources_sha16()),
        "stage_event_scale": kt.get("event_overhead_scale"),
        "bytes_per_launch": int(ab[dom]), "launch_us": round(launch_ms * 1e3, 2),
        "launches_per_mvm": (d + 1) if dom == "blur_axis" else 1,
        "note": ("cache-resident lattice (values + neighbour ids of an axis fit the L2s): a blur launch is a dependent-"
                 "launch boundary plus two memory latencies from 8 cold XCD L2s, not an HBM stream (DESIGN.md 4); bytes and "
                 "time are per axis (two-axes-per-launch kernels count twice); the HBM-bound regime is reported under 'fine'")
                if cache_resident else "",
    }
    # where the counters say the kernel moved FEWER bytes than the formula charges (the compacted blur skips ids it never
    # reads), the fraction on the bytes actually moved stands beside `frac`
    if pmc and pmc["bytes"] < ab[dom]:
        roof["achieved_on_traffic"] = round(pmc["bytes"] / (launch_ms * 1e-3) / 1e9, 1)
        roof["frac_on_traffic"] = round(pmc["bytes"] / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    stages = {}
    for k in per_mvm_ms:
        mult = (d + 1) if k == "blur_axis" else 1
        p = stage_traffic(k)
        stages[k] = {"us_per_mvm": round(per_mvm_ms[k] * 1e3, 2), "alg_MB_per_mvm": round(ab[k] * mult / 1e6, 2),
                     "GBps": round(ab[k] * mult / (per_mvm_ms[k] * 1e-3) / 1e9, 1),
                     "frac": round(ab[k] * mult / (per_mvm_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "kernels": names[k],
                     "traffic_MB_per_launch": round(p["bytes"] / 1e6, 1) if p else None}
        if p and p["bytes"] < ab[k]:
            launch_s = per_mvm_ms[k] * 1e-3 / mult
            stages[k]["frac_on_traffic"] = round(p["bytes"] / launch_s / 1e9 / HBM_PEAK_GBPS, 4)
    return roof, stages


def kernel_sources_sha16():
    """Fingerprint of the kernel sources (simplex_gp_amd/csrc/*.hip, *.h): a PMC table is only quoted for the sources it
    was collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "simplex_gp_amd", "csrc", "*.hip")) +
                       glob.glob(os.path.join(ROOT, "simplex_gp_amd", "csrc", "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_names, ell):
    """HBM bytes per launch of a stage's kernels (summed) from the newest committed PMC table (profiles/*_pmc.json,
    written by tools/summarize_profile.py from separate rocprofv3 --pmc passes of this same command), corrected as
    the microarch guide prescribes for gfx950 (FETCH_SIZE x2 + WRITE_SIZE).  None when a kernel is not in the table, or
    when the table was collected on other kernel sources than the ones in the tree (a stale table is not quoted)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            doc = json.load(open(path))
            if doc.get("kernel_sources_sha16") != kernel_sources_sha16():
                continue
            table = doc["by_lengthscale"].get(str(ell), {})
        except Exception:                      # noqa: BLE001
            continue
        total, found = 0.0, 0
        for want in kernel_names:
            for key, rec in table.items():
                name = key.rsplit("|", 1)[0]
                if name.startswith("plx::" + want) or name.startswith(want):
                    total += (2 * rec["fetch_KB"] + rec["write_KB"]) * 1024
                    found += 1
                    break
        if found == len(kernel_names) and found:
            return {"bytes": int(total), "source": os.path.basename(path)}
    return None


def cpu_baseline(x, v, ell):
    """One full-size MVM on the host: the reference's own extension if oracle/_ref
    is present (kind "reference"), else the C port (kind "port")."""
    import torch
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


class Job:
    """One sharded (or single-GPU) operator on synthetic rows: build / mvm closures for the timed loops."""

    def __init__(self, ctx, n_total, d, vd, ell, taps=RBF1):
        import torch
        from simplex_gp_amd.distributed import ShardedLatticeMVM, shard_bounds
        self.ctx, self.n_total, self.d, self.vd, self.ell, self.taps = ctx, n_total, d, vd, ell, taps
        self.lo, self.hi = shard_bounds(n_total, ctx.world, ctx.rank)
        # a rank generates and keeps only its own rows (the sharded build never needs the others)
        x, v = synth(n_total, d, vd, lo=self.lo, hi=self.hi)
        self.ref = (x / ell).contiguous().to(ctx.dev)
        self.v = v.contiguous().to(ctx.dev)
        self.out = torch.empty_like(self.v)
        self.x_cpu, self.v_cpu = x, v
        self.op = ShardedLatticeMVM.from_local_rows(self.ref, taps, n_total=n_total)
        self.lat = self.op.lattice

    def build(self):
        self.op.rebuild(self.ref, self.taps)

    def mvm(self):
        if self.ctx.dist is None:
            self.lat.apply(self.v, self.out)
        else:
            self.op.matmul(self.v, self.out)

    def rate(self, steps, warmup=3):
        """warm MVMs/s of the n_total operator (max over ranks of the wall time; the better of two timed regions: a leg that
        follows another one's allocations once measured 14x too slow for a single region -- round 5, config4 after config3).
        The headline `value` is NOT taken this way: it is exactly K steps in one region, as the contract says."""
        for _ in range(warmup):
            self.mvm()
        walls = [self.ctx.max_over_ranks(time_region(lambda i: self.mvm(), steps, self.ctx.sync, self.ctx.barrier)) for _ in range(2)]
        return steps / min(walls)

    def stage_us(self, reps=20):
        """Per-stage device time of the sharded MVM on this rank (torch events on the current stream), max over ranks."""
        import torch
        acc = {}
        for _ in range(reps):
            ev = []

            def mark(name, ev=ev):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))
            self.op.matmul(self.v, self.out, marks=mark)
            self.ctx.sync()
            for (_, a), (name, b) in zip(ev, ev[1:]):
                acc.setdefault(name, []).append(a.elapsed_time(b) * 1e3)
        return {k: round(self.ctx.max_over_ranks(float(np.mean(v))), 2) for k, v in acc.items()}

    def close(self):
        self.lat.close()


class Ctx:
    def __init__(self, args):
        import torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = torch.cuda.device_count()
        self.dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        self.dist = None
        self.backend = args.backend
        self.forced = self.world == 1 and os.environ.get("PLX_BENCH_SINGLE_RANK_RCCL") == "1"
        if self.world > 1 or self.forced:
            import torch.distributed as dist
            if self.forced:
                # one rank, real RCCL: every collective of the multi-rank path runs (as a no-op on the data)
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if "MASTER_PORT" not in os.environ:
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                from simplex_gp_amd import distributed as pd
                pd.FORCE_COLLECTIVES = True
                dist.init_process_group(args.backend, rank=0, world_size=1,
                                        **({"device_id": self.dev} if args.backend == "nccl" else {}))
            elif args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(args.backend)
            self.dist = dist

    def sync(self):
        import torch
        torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(device_ids=[self.dev_index]) if self.backend == "nccl" else self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], device=self.dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def graph_launches(fn):
    """What capturing fn() into a HIP graph records: {"kernels": kernel nodes, "nodes": all nodes (memsets and copies
    too)} -- the launch count of a piece of the hot loop without a profiler.  None when the capture is not possible
    (fn synchronises or allocates outside torch's graph pool)."""
    import ctypes
    import torch
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            fn()
        graph = ctypes.c_void_p(g.raw_cuda_graph())
        cnt = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(graph, None, ctypes.byref(cnt)) != 0:
            return None
        nodes = (ctypes.c_void_p * max(cnt.value, 1))()
        if hip.hipGraphGetNodes(graph, nodes, ctypes.byref(cnt)) != 0:
            return None
        kernels = 0
        for i in range(cnt.value):
            kind = ctypes.c_int(-1)
            hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(kind))
            kernels += kind.value == 0                      # hipGraphNodeTypeKernel
        return {"kernels": int(kernels), "nodes": int(cnt.value)}
    except Exception as e:                                   # noqa: BLE001  (a measuring aid: never fails the bench line)
        log(f"bench.py: graph_launches: {type(e).__name__}: {e}")
        return None


def cg_launch_count(model, x, rhs):
    """Launches per CG iteration of khat_solve on a built lattice: the difference between the captured graphs of a
    6- and a 2-iteration solve (tol = 0: no convergence read-back inside), divided by 4.  The operator K is made once,
    outside the captures: its scaled positions are the lattice-cache key, so the captured solves find the lattice built
    (a build reads counts back to the host and is refused under capture)."""
    import torch
    with torch.no_grad():
        K = model.kernel(x, x)
        torch.cuda.synchronize()
        for it in (2, 6):
            model.khat_solve(x, rhs, K=K, max_iter=it, tol=0.0)      # builds the lattice, sizes every buffer both captures ask for
        torch.cuda.synchronize()
        a = graph_launches(lambda: model.khat_solve(x, rhs, K=K, max_iter=2, tol=0.0))
        b = graph_launches(lambda: model.khat_solve(x, rhs, K=K, max_iter=6, tol=0.0)) if a else None
        torch.cuda.synchronize()
    if not a or not b:
        return None
    return {"kernels": (b["kernels"] - a["kernels"]) / 4.0, "graph_nodes": (b["nodes"] - a["nodes"]) / 4.0,
            "how": "HIP-graph capture of khat_solve at 6 and at 2 iterations, difference / 4 (kernel nodes; all nodes)"}


def cg_launch_leg(ctx, n=1_000_000, d=8):
    """config 3's CG iteration counted in launches.  Runs LAST: a capture that fails can leave the stream unusable, and by
    then every number of the line exists."""
    import torch
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    try:
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(n, d, generator=g).to(ctx.dev)
        rhs = torch.randn(n, 11, generator=g).to(ctx.dev)
        model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).to(ctx.dev)
        out = cg_launch_count(model, x, rhs)
        plx.lattice_cache().clear()
        return out
    except Exception as e:                                   # noqa: BLE001
        log(f"bench.py: cg_launch_leg: {type(e).__name__}: {e}")
        return None


def config3_leg(ctx, n=1_000_000, d=8, iters=50):
    """BASELINE.json configs[2] as the reference's training loop runs it (experiments/train_simplexgp.py:29-57):
    50 CG iterations on (s K + sigma^2 I) with right-hand side [y | 10 Rademacher probes] (vd = 11), GPyTorch default
    hyper-parameters (lengthscale = outputscale = softplus(0), noise softplus(0) + 1e-4).  One lattice build + 50
    MVMs; the lengthscale is nudged per trial so that every trial rebuilds."""
    import torch
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).to(ctx.dev)
    y = torch.randn(n, generator=g).to(ctx.dev)
    Z = (torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1).to(ctx.dev)
    rhs = torch.cat([y[:, None], Z], 1)
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).to(ctx.dev)
    best, best_warm, best_rebuild, res, m = float("inf"), float("inf"), float("inf"), None, None
    with torch.no_grad():
        for trial in range(4):
            # a COLD build: the point order is computed from the positions (the cache's warm start is switched off for this
            # solve; the lattice OBJECT is the recycled one, as in rounds 1-5: its buffers exist) -- the figure of rounds 1-5
            from simplex_gp_amd import lattice_kernel as lk
            saved_age, lk.MAX_ORDER_AGE = lk.MAX_ORDER_AGE, 0
            model.kernel.lengthscale = 0.6931 * (1 + 1e-5 * trial)
            ctx.sync()
            t0 = time.perf_counter()
            try:
                _, info = model.khat_solve(x, rhs, max_iter=iters, tol=0.0)
            finally:
                lk.MAX_ORDER_AGE = saved_age
            ctx.sync()
            dt = time.perf_counter() - t0
            t1 = time.perf_counter()
            _, info = model.khat_solve(x, rhs, max_iter=iters, tol=0.0)      # lattice cached: CG only
            ctx.sync()
            dtw = time.perf_counter() - t1
            # the rebuild a training step pays: the same data under a lengthscale that moved -- the cache rebuilds the
            # lattice in place with the point order kept (round 6: Lattice.build(reuse_order=True))
            model.kernel.lengthscale = 0.6931 * (1 + 1e-5 * trial + 5e-6)
            ctx.sync()
            t2 = time.perf_counter()
            _, info = model.khat_solve(x, rhs, max_iter=iters, tol=0.0)
            ctx.sync()
            dtr = time.perf_counter() - t2
            if trial > 0:
                best, best_warm, best_rebuild = min(best, dt), min(best_warm, dtw), min(best_rebuild, dtr)
            res = float(info["residual"].max())
            m = list(plx.lattice_cache()._entries.values())[-1][0].m
        # the rank-100 pivoted-Cholesky factor of the reference's recipe (train_simplexgp.py:36) on this lattice, by itself
        pre = model.preconditioner(x, 100)
        fbest = float("inf")
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            p2 = solvers.LatticePreconditioner(pre.lat, float(model.outputscale), float(model.noise), 100)
            ctx.sync()
            fbest = min(fbest, time.perf_counter() - t0)
        factor = {"ms": round(fbest * 1e3, 2), "rank": 100, "batches": p2.batches, "planned_batches": p2.planned_batches,
                  "frontier_batches": p2.sparse_batches, "m_vertices": p2.lat.m,
                  "what": "solvers.LatticePreconditioner on the built lattice: kernel rows (plx_filter_onehot), batches "
                          "(plx_pchol_*), fp16 rounding, Gram matrix and its host Cholesky"}
        del pre, p2
    # the evaluation the loop runs beside the step (train_simplexgp.py:123-165) at this size: one training.predict with the
    # recipe's settings (cg_eval_tol 1e-2, 100 Lanczos steps, rank-100 preconditioner) on n / 4 held-out rows (the
    # reference's 64 / 16 / 20 split: validation = train / 4), and its cache (solve + Lanczos run) by itself
    from simplex_gp_amd import training
    gs = torch.Generator().manual_seed(4321)
    xs = torch.randn(n // 4, d, generator=gs).to(ctx.dev)
    model.kernel.lengthscale = 0.6931
    ys = torch.sin(x[:, 0]) + 0.1 * y
    training.predict(model, x, ys, xs, cg_tol=1e-2, lanc_iter=100, pre_size=100)
    # the state of the loop's next epoch: the same data, the lengthscale moved -- both lattices (x; [x*; x]) are rebuilt in
    # place with their point order kept, nothing is allocated
    model.kernel.lengthscale = 0.6931 * (1 + 1e-5)
    ctx.sync()
    t0 = time.perf_counter()
    cache = training.PredictionCache(model, x, ys, cg_tol=1e-2, lanc_iter=100, pre_size=100)
    ctx.sync()
    t1 = time.perf_counter()
    cache.predict(xs)
    ctx.sync()
    evaluation = {"cache_ms": round((t1 - t0) * 1e3, 2), "split_ms": round((time.perf_counter() - t1) * 1e3, 2),
                  "cg_iterations": int(cache.solve_info.get("iterations", -1)), "held_out_rows": n // 4,
                  "what": "training.PredictionCache (rank-100 factor on the lattice already built + preconditioned CG to 1e-2 + 100 "
                          "Lanczos steps through plx_lanczos_step) and one split's mean + variance through the rectangular "
                          "operator (101 columns), lattice rebuilds included (the lengthscale moved since the previous "
                          "evaluation); the loop pays the cache once and the split twice per epoch"}
    del cache, xs, ys
    plx.lattice_cache().clear()
    del x, y, Z, rhs
    train = train_step_leg(ctx, n, d, lambda: plx.RBFLattice(order=1, ard_num_dims=d))
    return {"config3_cg_ms": round(best * 1e3, 2), "config3": {
        "workload": f"N={n}, d={d}, vd=11, lengthscale 0.6931, {iters} CG iterations incl. one lattice build",
        "ms_incl_build": round(best * 1e3, 2), "ms_cg_only": round(best_warm * 1e3, 2),
        "ms_incl_warm_rebuild": round(best_rebuild * 1e3, 2),       # the lattice of a moved lengthscale, point order kept
        "m_vertices": m,
        "cg_iterations_per_s": round(iters / best_warm, 1), "final_rel_residual_max": res, "factor": factor,
        "evaluation": evaluation,
        "launches_per_cg_iteration": None,         # filled in by main() as the very last measurement (cg_launch_leg)
        "train_step_ms": {k: v["step_ms"] for k, v in train.items() if k.startswith("pre_size")}, "train_step": train}}


def train_step_leg(ctx, n, d, make_kernel, pre_sizes=(0, 100), steps=3, min_noise=1e-4, label=""):
    """One marginal-likelihood training step as the reference's loop runs it (experiments/train_simplexgp.py:29-57:
    10 probes, cg_tolerance(1.0), max_cg_iterations(500), max_preconditioner_size(pre_size); loss.backward(); Adam):
    wall time of the whole step (forward + backward + optimiser, no synchronisation inside; best of `steps` after one
    untimed step) and, from one more step run in the profiling mode of solvers.marginal_log_likelihood (a device
    synchronisation at every phase boundary), where it goes."""
    import torch
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).to(ctx.dev)
    y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).to(ctx.dev))
    out = {}
    for pre in pre_sizes:
        model = solvers.LatticeGP(make_kernel(), min_noise=min_noise).to(ctx.dev)
        from simplex_gp_amd import training
        opt = training.make_optimizer(model, lr=0.1)
        start = {k: v.detach().clone() for k, v in model.state_dict().items()}

        def step(seed, prof=None):
            # every measured step starts from the same (GPyTorch default) hyper-parameters: the lattice of a step depends
            # on the lengthscale, and Adam at lr 0.1 moves it by 10 % per step
            model.load_state_dict(start)
            opt.zero_grad()
            ctx.sync()
            t0 = time.perf_counter()
            mll = solvers.marginal_log_likelihood(model, x, y, num_probes=10, cg_tol=1.0, max_cg_iter=500, seed=seed,
                                                  pre_size=pre, profile=prof)
            if prof is not None:
                ctx.sync()
            t1 = time.perf_counter()
            (-mll).backward()
            if prof is not None:
                ctx.sync()
                prof["backward"] = (time.perf_counter() - t1) * 1e3
            t2 = time.perf_counter()
            opt.step()
            ctx.sync()
            if prof is not None:
                prof["optimizer"] = (time.perf_counter() - t2) * 1e3
            return (time.perf_counter() - t0) * 1e3, mll
        step(0)
        step(0)                                   # two untimed steps: the lattice objects of forward and backward exist and are sized
        best = min(step(1 + i)[0] for i in range(steps))
        # phases: the MEDIAN over three profiled steps (a single profiled step once reported an "optimizer" phase of 73 ms
        # inside a 12.8 ms step: an outlier of that one step, not a property of the phase); `phases_sum_ms` is what they add up
        # to -- a synchronised step, so somewhat above step_ms
        profs, prof_walls = [], []
        for i in range(3):
            pr = {}
            wall_i, mll = step(99 + i, pr)
            profs.append(pr)
            prof_walls.append(wall_i)
        keys = [k for k in profs[0]]
        phases = {k: float(np.median([pr.get(k, 0.0) for pr in profs])) for k in keys}
        lat = list(plx.lattice_cache()._entries.values())[-1][0]
        out[f"pre_size_{pre}"] = {"step_ms": round(best, 2), "phases_ms": {k: round(v, 2) for k, v in phases.items()},
                                  "phases_sum_ms": round(sum(phases.values()), 2), "phases_from": "median of 3 profiled steps",
                                  "profiled_step_ms": round(float(np.median(prof_walls)), 2),
                                  "cg_iterations": int(mll.cg_info["iterations"]), "m_vertices": lat.m}
        plx.lattice_cache().clear()
        del model, opt
    out["workload"] = (f"{label}N={n}, d={d}: one Adam step on the CG/SLQ marginal likelihood (10 probes, cg_tol 1, max 500 CG "
                       "iterations, GPyTorch's default initial hyper-parameters): forward (preconditioner, probes, solve, SLQ, "
                       "differentiable MVM) + backward (one 2L(1+d)-column filter with the derivative taps) + optimiser")
    return out


def config5_leg(ctx, n=10623, d=18):
    """BASELINE.json configs[4] stand-in (UCI elevators is not available offline): MaternLattice(nu=1.5, order=3)
    at N = 0.64 x 16,599 training points, d = 18: one warm MVM and one cold (build + apply) call."""
    import torch
    import simplex_gp_amd as plx
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).to(ctx.dev)
    v = torch.randn(n, 1, generator=g).to(ctx.dev)
    k = plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d)
    taps = k.dkernel_fn.get_coeffs().numpy()
    lat = plx.Lattice(ctx.dev)
    lat.build(x, taps)
    out = torch.empty_like(v)
    for _ in range(5):
        lat.apply(v, out)
    reps = 50
    wall = time_region(lambda i: lat.apply(v, out), reps, ctx.sync, lambda: None)
    wall_cold = time_region(lambda i: (lat.build(x, taps), lat.apply(v, out)), 10, ctx.sync, lambda: None)
    m = lat.m
    lat.close()
    # one epoch of the reference's recipe for this data set (configs/simplexgp.yml:11-45: pre_size 100, min_noise 0.1,
    # Matern-1.5; order 3 per BASELINE.json): a training step, and the evaluation the loop runs beside it
    # (train_simplexgp.py:123-165: predictive mean + variance on the validation and test splits, cg_eval_tol 1e-2)
    train = train_step_leg(ctx, n, d, lambda: plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d), pre_sizes=(100,), steps=3,
                           min_noise=0.1, label="elevators stand-in, ")
    from simplex_gp_amd import solvers, training
    xs = torch.randn(2656 + 3320, d, generator=g).to(ctx.dev)                 # 16 % + 20 % of 16,599 rows
    y = (torch.sin(x[:, 0]) + 0.1 * torch.randn(n, generator=g).to(ctx.dev))
    model = solvers.LatticeGP(plx.MaternLattice(nu=1.5, order=3, ard_num_dims=d), min_noise=0.1).to(ctx.dev)
    training.predict(model, x, y, xs, cg_tol=1e-2, lanc_iter=100, pre_size=100)
    ctx.sync()
    t0 = time.perf_counter()
    training.predict(model, x, y, xs, cg_tol=1e-2, lanc_iter=100, pre_size=100)
    ctx.sync()
    eval_ms = (time.perf_counter() - t0) * 1e3
    # the loop as the reference runs it (train_simplexgp.py:123-165, log_int = 1): a step, then the evaluation of the
    # validation AND the test split (one mean / variance cache for both, as GPyTorch's eval mode keeps it; the next step
    # finds the evaluation's lattice and preconditioner: the optimiser has not moved in between)
    xv, xt = xs[:2656], xs[2656:]
    yv, yt = torch.sin(xv[:, 0]), torch.sin(xt[:, 0])
    loop_epochs = 6
    stamps = []

    def stamp(row):
        ctx.sync()
        stamps.append(time.perf_counter())
    training.fit(model, (x, y), val=(xv, yv), test=(xt, yt), epochs=loop_epochs, lr=0.1, num_probes=10, cg_iter=500, cg_tol=1.0,
                 cg_eval_tol=1e-2, lanc_iter=100, pre_size=100, log=stamp)
    loop_ms = (stamps[-1] - stamps[1]) / (loop_epochs - 2) * 1e3                # epochs 3..6 (the first two size buffers)
    plx.lattice_cache().clear()
    return {"config5_mvm_us": round(wall / reps * 1e6, 1), "config5": {
        "workload": f"MaternLattice(nu=1.5, order=3) stand-in for elevators: N={n}, d={d}, vd=1, lengthscale 1",
        "m_vertices": m, "warm_mvm_us": round(wall / reps * 1e6, 1), "cold_call_us": round(wall_cold / 10 * 1e6, 1),
        "epoch_ms": train["pre_size_100"]["step_ms"], "train_step": train, "eval_ms": round(eval_ms, 2),
        "eval_workload": "training.predict (CG mean at cg_eval_tol 1e-2 + 100-step Lanczos variance, pre_size 100) on "
                         "5,976 held-out rows",
        "loop_epoch_ms": round(loop_ms, 2),
        "loop_workload": "training.fit, mean of epochs 3-6: one step + the evaluation of a 2,656-row validation and a 3,320-row "
                         "test split (RMSE / MAE / NLL each; one solve + one Lanczos run for both), as "
                         "experiments/train_simplexgp.py:123-165 runs every epoch"}}


# The only MVM timings the reference publishes (notebooks/viz_compute.ipynb:102-106: `simplex_mvm_t`, seconds per MVM of its
# CUDA path on the UCI sets; GPU not stated).  The data sets are not redistributable: the shapes are run on synthetic
# standardised clouds.
PUBLISHED_SHAPES = [("houseelectric", 2_049_280, 11, 1.756), ("precipitation", 628_474, 3, 0.082), ("keggdirected", 48_827, 20, 0.134),
                    ("elevators", 16_599, 17, 0.083), ("protein", 45_730, 9, 0.034)]


def published_shapes_leg(ctx, ell=0.6931):
    """One K.v MVM (vd = 1, RBF order 1, lengthscale softplus(0) = GPyTorch's initial value) at the (n, d) of the reference's
    published timing table, on x ~ N(0, I) seed 1234: warm (lattice built) and cold (plx_filter: build + apply, what the
    reference's filter() does on every call) seconds, beside the published CUDA seconds -- different data, unstated GPU:
    an order-of-magnitude placement, not a like-for-like ratio."""
    import torch
    import simplex_gp_amd as plx
    rows = []
    for name, n, d, pub in PUBLISHED_SHAPES:
        g = torch.Generator().manual_seed(1234)
        ref = (torch.randn(n, d, generator=g) / ell).contiguous().to(ctx.dev)
        v = torch.randn(n, 1, generator=g).to(ctx.dev)
        out = torch.empty_like(v)
        lat = plx.Lattice(ctx.dev)
        lat.build(ref, RBF1)
        lat.prepare(1)
        for _ in range(3):
            lat.apply(v, out)
        reps = 20
        warm = time_region(lambda i: lat.apply(v, out), reps, ctx.sync, lambda: None) / reps
        lat.filter_once(v, ref, RBF1, out)
        ncold = 5
        cold = time_region(lambda i: lat.filter_once(v, ref, RBF1, out), ncold, ctx.sync, lambda: None) / ncold
        rows.append({"shape_of": name, "n": n, "d": d, "m_vertices": lat.m, "warm_mvm_s": round(warm, 7), "cold_mvm_s": round(cold, 6),
                     "published_cuda_mvm_s": pub, "published_over_cold": round(pub / cold, 1)})
        lat.close()
        del ref, v, out
    return {"published_shapes": rows,
            "published_shapes_note": "reference: notebooks/viz_compute.ipynb:102-106 (its CUDA path, one filter() = build + apply "
                                     "per MVM, real UCI data, GPU not stated); here: synthetic N(0, I) clouds of the same (n, d), "
                                     "lengthscale 0.6931, vd = 1 -- different data, so a placement, not a like-for-like ratio"}


def exchange_record(job, vd, stage_us):
    """The one exchange step of a sharded MVM: what moves, how many bytes per rank, and its device time."""
    transport = "RCCL" if job.ctx.backend == "nccl" else job.ctx.backend
    return {"kind": "all_reduce(sum) of the vertex accumulator values[m, %d] (fp32), %s" % (job.lat.values_stride(vd), transport),
            "bytes": job.op.exchange_bytes(vd), "us": stage_us.get("exchange")}


def expected_speedup(stage_us, world, build=None):
    """What a leg's own measured stage times predict for "this operator on `world` GPUs against ONE GPU", so that a
    SCALE record reads as model confirmed / refuted rather than as a bare curve.  The model is DESIGN.md 5's: splat and
    slice shard by points (one GPU does `world` times this rank's share -- linear in the rows: where a stage is latency-
    rather than byte-bound, i.e. on coarse lattices, one GPU needs LESS than that, so the figure is an upper bound on
    the speedup), the blur is replicated (the same time on one GPU), the exchange exists only between ranks.
    build = (this rank's build ms with the key exchange, MVMs per build): the bench cadence's build share, modelled as
    not sharded at all (the merge works on every rank's keys)."""
    sp, ex, bl, sl = (float(stage_us.get(k) or 0.0) for k in ("splat", "exchange", "blur", "slice"))
    t_rank = sp + ex + bl + sl
    if t_rank <= 0.0:
        return None
    t_one = world * (sp + sl) + bl
    out = {"model": "T_1 = world x (splat + slice) + blur;  T_world = splat + exchange + blur + slice  (this rank's stage_us)",
           "t_one_gpu_us": round(t_one, 1), "t_rank_us": round(t_rank, 1), "warm_mvm": round(t_one / t_rank, 2),
           "warm_mvm_if_exchange_were_free": round(t_one / max(t_rank - ex, 1e-9), 2),
           "replicated_share_of_rank_time": round((bl + ex) / t_rank, 3)}
    if build is not None:
        b_us, per = build[0] * 1e3, max(int(build[1]), 1)
        out["with_one_build_per_%d_mvms" % per] = round((t_one + b_us / per) / (t_rank + b_us / per), 2)
    return out


def sharded_leg(ctx, n_total, d, ell, vds, steps):
    """A fixed-total-size operator sharded over the ranks (strong scaling / config 4): plain MVMs/s per vd, per-stage
    device time per rank (max over ranks) and the exchange."""
    multi = ctx.dist is not None
    out = {"n_total": n_total, "lengthscale": ell, "rccl_ranks": ctx.world}
    for vd in vds:
        job = Job(ctx, n_total, d, vd, ell)
        out["m_vertices"] = job.op.m
        out[f"mvms_per_s_vd{vd}"] = round(job.rate(steps), 1)
        if multi:
            st = job.stage_us(10)
            out[f"stage_us_vd{vd}"] = st
            out[f"exchange_vd{vd}"] = exchange_record(job, vd, st)
            out[f"expected_speedup_vs_1gpu_vd{vd}"] = expected_speedup(st, ctx.world)
        elif vd > 1:
            # one GPU: the same operator driven the way a CG solve drives it -- rows in lattice order, columns padded to
            # whole 16-byte vectors (solvers.khat_solve); the caller-order figure above pays two row permutations per MVM
            import torch
            lat = job.lat
            lat.set_lattice_row_order(True)
            vdp = lat.values_stride(vd)
            v_l = torch.zeros((job.v.shape[0], vdp), device=job.v.device)
            v_l[:, :vd] = lat.to_lattice_order(job.v)
            out_l = torch.empty_like(v_l)
            for _ in range(3):
                lat.apply(v_l, out_l)
            wall = time_region(lambda i: lat.apply(v_l, out_l), steps, ctx.sync, ctx.barrier)
            out[f"mvms_per_s_vd{vd}_lattice_rows"] = round(steps / wall, 1)
            lat.set_lattice_row_order(False)
        job.close()
        del job
    return out


def timed_collective(ctx, fn, reps=10):
    """Mean wall time (us) of a collective issued alone, max over ranks."""
    for _ in range(2):
        fn()
    wall = time_region(lambda i: fn(), reps, ctx.sync, ctx.barrier)
    return round(ctx.max_over_ranks(wall) / reps * 1e6, 1)


def config3_multi_leg(ctx, n=1_000_000, d=8, iters=50):
    """BASELINE.json configs[2] on several GPUs, both ways of splitting ONE batched solve (N = 1e6, lengthscale 0.6931,
    [y | 10 probes], 50 CG iterations incl. the lattice build):
      points   rows sharded, one vertex all-reduce per MVM + the dot-product all-reduces (distributed.sharded_solve)
      columns  every rank builds the whole lattice and solves its share of the 11 columns with the single-GPU solver
               (lattice row order, fused kernels); no collective inside the iteration, one all-gather of the solution"""
    import torch
    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    from simplex_gp_amd.distributed import (ShardedLatticeMVM, shard_bounds, sharded_solve, column_sharded_solve, column_bounds,
                                            all_gather_columns, all_reduce_sum)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g)
    y = torch.randn(n, generator=g)
    Z = torch.randint(0, 2, (n, 10), generator=g).float() * 2 - 1
    rhs = torch.cat([y[:, None], Z], 1)
    t = rhs.shape[1]
    ell, s, noise = 0.6931, 0.6931, 0.6932
    transport = "RCCL" if ctx.backend == "nccl" else ctx.backend
    out = {"workload": f"N={n}, d={d}, vd={t}, lengthscale {ell}, {iters} CG iterations incl. one lattice build, {ctx.world} rank(s)"}
    # ---- rows sharded
    lo, hi = shard_bounds(n, ctx.world, ctx.rank)
    rhs_local = rhs[lo:hi].contiguous().to(ctx.dev)
    op = ShardedLatticeMVM.from_local_rows((x[lo:hi] / ell).contiguous().to(ctx.dev), RBF1, n_total=n)
    sharded_solve(op, rhs_local, s, noise, max_iter=3, tol=0.0)
    best = float("inf")
    for trial in range(3):
        ref_local = (x[lo:hi] / (ell * (1 + 1e-5 * trial))).contiguous().to(ctx.dev)
        ctx.barrier(); ctx.sync()
        t0 = time.perf_counter()
        op.rebuild(ref_local, RBF1)
        _, info = sharded_solve(op, rhs_local, s, noise, max_iter=iters, tol=0.0)
        ctx.sync()
        best = min(best, ctx.max_over_ranks(time.perf_counter() - t0))
    vals = op.lattice.new_values(t)
    out["points"] = {"ms_incl_build": round(best * 1e3, 2), "m_vertices": op.m, "rows_per_rank": hi - lo,
                     "final_rel_residual_max": float(info["residual"].max()),
                     "exchange": {"kind": f"per MVM: all_reduce(sum) of values[m, {op.lattice.values_stride(t)}] (fp32); per iteration: "
                                          f"2 all_reduce of {t} dot products; {transport}",
                                  "bytes": op.exchange_bytes(t), "us": timed_collective(ctx, lambda: all_reduce_sum(vals, op.group))}}
    op.lattice.close()
    del op, vals, rhs_local
    # ---- columns sharded
    xd, rd = x.to(ctx.dev), rhs.to(ctx.dev)
    model = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).to(ctx.dev)
    clo, chi = column_bounds(t, ctx.world, ctx.rank)
    best = float("inf")
    with torch.no_grad():
        for trial in range(4):
            model.kernel.lengthscale = ell * (1 + 1e-5 * trial)
            ctx.barrier(); ctx.sync()
            t0 = time.perf_counter()
            X, info = column_sharded_solve(lambda B: model.khat_solve(xd, B, max_iter=iters, tol=0.0), rd)
            ctx.sync()
            dt = ctx.max_over_ranks(time.perf_counter() - t0)
            if trial > 0:
                best = min(best, dt)
    blk = X[:, clo:chi].contiguous()
    out["columns"] = {"ms_incl_build": round(best * 1e3, 2), "columns_on_rank0": [clo, chi], "max_columns_per_rank": -(-t // ctx.world),
                      "final_rel_residual_max": float(info["residual"].max()),
                      "exchange": {"kind": f"once per solve: all_gather of the solution's column blocks [n, {t}] (fp32); nothing inside "
                                           f"the iteration; {transport}",
                                   "bytes": n * t * 4, "us": timed_collective(ctx, lambda: all_gather_columns(blk, t))}}
    plx.lattice_cache().clear()
    # ---- the training step, column-sharded (distributed.column_sharded_mll): same recipe as the one-GPU `config3.train_step`
    from simplex_gp_amd.distributed import column_sharded_mll, all_reduce_gradients
    yv = (torch.sin(xd[:, 0]) + 0.1 * torch.randn(n, generator=torch.Generator().manual_seed(99)).to(ctx.dev))
    steps_out = {}
    for pre in (0, 100):
        m2 = solvers.LatticeGP(plx.RBFLattice(order=1, ard_num_dims=d)).to(ctx.dev)
        opt = torch.optim.Adam(m2.parameters(), lr=0.1)
        start = {k: v.detach().clone() for k, v in m2.state_dict().items()}

        def step(seed):
            m2.load_state_dict(start)
            opt.zero_grad()
            ctx.barrier(); ctx.sync()
            t0 = time.perf_counter()
            val = column_sharded_mll(m2, xd, yv, num_probes=10, cg_tol=1.0, max_cg_iter=500, seed=seed, pre_size=pre)
            (-val).backward()
            all_reduce_gradients(m2)
            opt.step()
            ctx.sync()
            return ctx.max_over_ranks(time.perf_counter() - t0), float(val.detach())
        step(0); step(0)
        best, val = min(step(1 + i) for i in range(3))
        steps_out[f"pre_size_{pre}"] = {"step_ms": round(best * 1e3, 2), "mll": round(val, 5)}
        plx.lattice_cache().clear()
    out["train_step_columns"] = dict(steps_out, exchange={"kind": f"per step: all_reduce of 2 scalars (value) + all_reduce of the {d + 3} "
                                                                     f"hyper-parameter gradients; {transport}", "bytes": 8 * 2 + 4 * (d + 3)},
                                     workload="one Adam step on the CG/SLQ marginal likelihood, [y | 10 probes] columns sharded over the "
                                              "ranks, replicated lattice build and preconditioner")
    return out


CONFIG4_POINTS = int(os.environ.get("PLX_BENCH_CONFIG4_POINTS", "4000000"))      # (tests rehearse the multi-rank legs at a smaller size)


def columns_leg(ctx, n, d, ell, steps):
    """The metric's operator (N = 1e6, d = 8, vd = 1) in COLUMNS mode: every rank builds the whole lattice and applies it to its
    own right-hand-side column -- the way the columns of a batched solve are sharded (distributed.column_sharded_solve): no
    collective on the data path at all.  Rate = single-column K.v MVMs per second over ALL ranks (every one of them a whole
    N-point MVM); per-GPU work is fixed as the rank count grows, i.e. this is weak scaling in the number of right-hand sides,
    reported beside `value` (one MVM's rows sharded over the ranks), never instead of it."""
    import torch
    import simplex_gp_amd as plx
    x, v = synth(n, d, 1, seed=1234 + 17 * ctx.rank)                 # (the same positions matter, not the same vectors)
    x0, _ = synth(n, d, 1)
    ref = (x0 / ell).contiguous().to(ctx.dev)
    vv = v.contiguous().to(ctx.dev)
    out = torch.empty_like(vv)
    lat = plx.Lattice(ctx.dev)
    lat.build(ref, RBF1)
    lat.prepare(1)
    for _ in range(20):
        lat.apply(vv, out)

    def step(i):
        if i % 20 == 0:
            lat.build(ref, RBF1)
        lat.apply(vv, out)
    step(0)
    wall_warm = ctx.max_over_ranks(time_region(lambda i: lat.apply(vv, out), steps, ctx.sync, ctx.barrier))
    wall_cad = ctx.max_over_ranks(time_region(step, steps, ctx.sync, ctx.barrier))
    lat.close()
    return {"mvms_per_s_all_ranks_warm": round(ctx.world * steps / wall_warm, 1),
            "mvms_per_s_all_ranks_one_build_per_20": round(ctx.world * steps / wall_cad, 1), "ranks": ctx.world,
            "scaling": "weak (one right-hand-side column per rank; every MVM is a whole N-point K.v on one GPU)",
            "exchange": {"kind": "none", "bytes": 0, "us": 0.0}}


def config4_grid_leg(ctx, steps, n=CONFIG4_POINTS, d=8, vd=11):
    """BASELINE.json configs[3] at vd = 11 on the points x columns grid (distributed.SolveGrid): C column groups, each a
    row-sharded operator over P = world / C ranks.  Every factorisation of the rank count is timed; the rate is 11-column
    MVMs per second = 1 / (the slowest rank's time for its column block)."""
    import torch
    from simplex_gp_amd.distributed import ShardedLatticeMVM, SolveGrid
    out = {"n_total": n, "vd": vd, "lengthscale": 1.0}
    transport = "RCCL" if ctx.backend == "nccl" else ctx.backend
    for C in [c for c in (1, 2, 4, 8, 16) if ctx.world % c == 0 and c <= min(ctx.world, vd)]:
        grid = SolveGrid(C)
        lo, hi = grid.rows(n)
        clo, chi = grid.columns(vd)
        x, v = synth(n, d, vd, lo=lo, hi=hi)
        op = ShardedLatticeMVM.from_local_rows(x.contiguous().to(ctx.dev), RBF1, group=grid.point_group, n_total=n)
        vb = v[:, clo:chi].contiguous().to(ctx.dev)
        outb = torch.empty_like(vb)
        for _ in range(3):
            op.matmul(vb, outb)
        wall = ctx.max_over_ranks(time_region(lambda i: op.matmul(vb, outb), steps, ctx.sync, ctx.barrier))
        cols = chi - clo
        out[f"C{C}xP{grid.P}"] = {"mvms_per_s": round(steps / wall, 1), "columns_per_rank_max": -(-vd // C), "rows_per_rank": hi - lo,
                                  "m_vertices": op.m,
                                  "exchange": {"kind": ("none (P = 1: every rank holds its column block of the whole operator)" if grid.P == 1 else
                                                        f"per MVM: all_reduce(sum) of values[m, {op.lattice.values_stride(cols)}] inside the column "
                                                        f"group's {grid.P} ranks; {transport}"),
                                               "bytes": 0 if grid.P == 1 else op.exchange_bytes(cols)}}
        op.lattice.close()
        del op, vb, outb
    best = max((k for k in out if k.startswith("C")), key=lambda k: out[k]["mvms_per_s"])
    out["best"] = best
    return out


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args))
    if env_world is not None and int(env_world) != args.gpus:
        log(f"bench.py: WORLD_SIZE={env_world} but --gpus {args.gpus}: launch N ranks for --gpus N")
        sys.exit(2)

    import torch
    ctx = Ctx(args)
    world, rank, dist = ctx.world, ctx.rank, ctx.dist

    import simplex_gp_amd as plx
    from simplex_gp_amd import solvers
    # the host-side BLAS pool under the container's CPU quota (every rank of a multi-GPU run: 8 x 128 threads otherwise)
    host_threads = solvers.cap_host_threads()

    d, vd, r = args.d, args.vd, 1
    multi = ctx.dist is not None                 # several ranks (or the single-rank RCCL rehearsal): the sharded code path
    if args.scaling == "weak":
        n_total = args.n * world
    elif args.scaling == "strong":
        n_total = args.n
    else:
        n_total = 4_000_000
    job = Job(ctx, n_total, d, vd, args.ell)
    lat, m = job.lat, job.op.m
    n_local = job.hi - job.lo

    def step(i):
        if i % args.rebuild_every == 0:
            job.build()
        job.mvm()

    if args.dump:
        # rows of one MVM for an external checker (tests/check_bench_dump.py compares them with the CPU oracle);
        # bench.py itself only touches oracle/ in the cpu_baseline leg
        job.mvm()
        ctx.sync()
        np.savez(f"{args.dump}.rank{rank}.npz", out=job.out.cpu().numpy(), lo=job.lo, hi=job.hi, n_total=n_total, d=d,
                 vd=vd, ell=args.ell, m=m)

    # clocks first: a GPU that has idled through data generation starts the first milliseconds below its steady clocks
    # (measured: the same 20-step region 155 us / step cold, 143 us after 20 ms of MVMs); the requested warm-up steps
    # (which include a build) follow
    job.build()
    for _ in range(PREWARM_MVMS):
        job.mvm()
    for i in range(args.warmup):
        step(i)
    wall = ctx.max_over_ranks(time_region(step, args.steps, ctx.sync, ctx.barrier))
    builds = len([i for i in range(args.steps) if i % args.rebuild_every == 0])
    mvms_per_s = args.steps / wall
    value = mvms_per_s                            # always the plain rate of the n_total-point operator
    scaling = "weak" if (args.scaling == "weak" and world > 1) else "strong"

    result = {
        "metric": "lattice K.v MVMs/sec, N=1e6 d=8 order=1" + (" (1e6 points per GPU)" if scaling == "weak" else
                                                              (" (N=4e6: configs[3])" if args.scaling == "config4" else "")),
        "value": round(value, 2), "unit": "MVMs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"N={n_total} points over {world} GPU(s) ({n_local} on rank 0), d={d}, vd={vd}, RBFLattice "
                        f"order=1, lengthscale {args.ell}, x~N(0,I) seed 1234; timed region = {builds} lattice "
                        f"build(s) + {args.steps} MVMs (rebuild every {args.rebuild_every} MVMs: the CG loop of "
                        f"BASELINE.json configs[2])",
            "n_total": n_total, "m_vertices": m, "rebuild_every": args.rebuild_every,
            "builds_in_timed_region": builds, "scaling_mode": args.scaling, "prewarm_mvms": PREWARM_MVMS,
            "parallelism": "single GPU" if world == 1 else f"points sharded x{world}; build: local + all-gather of vertex "
                                                              "keys + merge; MVM: RCCL all-reduce of vertex values, replicated blur",
            "value_definition": "MVMs/s of the n_total-point operator (never scaled by the problem size)",
        },
        "mvms_per_s": round(mvms_per_s, 2),
        "host_threads": host_threads,      # torch's BLAS pool after solvers.cap_host_threads() (half the cgroup CPU quota)
    }

    if multi:
        # ---- what the ranks spent where, the exchange, and the other scaling curves
        leg_wall = {"main": round(time.perf_counter() - T_START, 2)}       # wall seconds per leg on rank 0 (import + data + timed region so far)

        def timed_leg(name, fn):
            t0 = time.perf_counter()
            val = fn()
            ctx.sync()
            leg_wall[name] = round(time.perf_counter() - t0, 2)
            log(f"bench.py: leg {name}: {leg_wall[name]} s")
            return val
        result["rccl_ranks"] = dist.get_world_size()
        result["backend"] = args.backend
        result["warm_mvms_per_s"] = round(job.rate(args.steps), 1)
        result["stage_us"] = job.stage_us(20)
        result["exchange"] = exchange_record(job, vd, result["stage_us"])
        result["allreduce_bytes"] = job.op.exchange_bytes(vd)
        result["build_key_bytes_exchanged"] = getattr(job.op, "key_bytes_exchanged", None)
        # the build's one exchange, timed (device synchronised around it): the all-gather of the per-rank vertex keys
        from simplex_gp_amd import distributed as pdist
        pdist.TIME_GATHER = True
        job.build()
        ctx.sync()
        pdist.TIME_GATHER = False
        result["build_key_allgather"] = {"kind": "all_gather_into_tensor of the per-rank vertex keys (padded to the largest rank) + "
                                                 "one small all_gather of the counts",
                                         "bytes": pdist.LAST_GATHER.get("bytes_out"), "us": round(pdist.LAST_GATHER.get("us") or 0.0, 1)}
        # the prediction next to the measurement: what these stage times say `value` should be against one GPU (the
        # cadence's build share from the difference between the timed region and its warm MVMs)
        build_ms_rank = max(0.0, (wall - args.steps / result["warm_mvms_per_s"]) / max(builds, 1) * 1e3) if builds else 0.0
        result["expected_speedup_vs_1gpu"] = expected_speedup(result["stage_us"], world,
                                                              build=(build_ms_rank, args.steps // max(builds, 1)) if builds else None)
        result["expected_speedup_note"] = ("no BASELINE config is projected to reach the north star's 6x at 8 GPUs: the blur is "
                                           "replicated and one all-reduce of values[m, vd] sits between splat and blur "
                                           "(DESIGN.md 5); the driver's curve is to be read against these figures")
        job.close()
        if not args.no_configs:
            short = max(10, args.steps // 2)
            if args.scaling != "strong":
                result["strong"] = timed_leg("strong", lambda: sharded_leg(ctx, args.n, d, args.ell, [1], short))
            if args.scaling != "config4":
                result["config4"] = timed_leg("config4", lambda: sharded_leg(ctx, CONFIG4_POINTS, d, 1.0, [1, 11], short))
            if args.scaling != "weak":
                result["weak_1e6_per_gpu"] = timed_leg("weak_1e6_per_gpu", lambda: sharded_leg(ctx, args.n * world, d, args.ell, [1], short))
            # weak scaling with 4e6 points per GPU: where the sharded splat / slice outweigh the replicated blur and the
            # all-reduce
            if not args.no_weak4:
                result["weak_4e6_per_gpu"] = timed_leg("weak_4e6_per_gpu", lambda: sharded_leg(ctx, CONFIG4_POINTS * world, d, 1.0, [1], short))
            result["columns_mode"] = timed_leg("columns_mode", lambda: columns_leg(ctx, args.n, d, args.ell, max(20, args.steps)))
            # one batched solve split by rows or by columns, and config 4 at vd = 11 on the points x columns grid
            result["config3_cg"] = timed_leg("config3_cg", lambda: config3_multi_leg(ctx, n=args.n))
            cfg4 = result["config4"] if args.scaling != "config4" else result.setdefault("config4", {})
            cfg4["grid_vd11"] = timed_leg("config4_grid_vd11", lambda: config4_grid_leg(ctx, short))
        result["leg_wall_s"] = leg_wall
        result["total_wall_s"] = round(time.perf_counter() - T_START, 2)
    else:
        # ---- warm / cold rates and per-stage times on the same lattice
        ref, v, out = job.ref, job.v, job.out
        wall_warm = time_region(lambda i: job.mvm(), args.steps, ctx.sync, ctx.barrier)
        ncold = max(5, args.steps // 5)
        # cold = the reference's one-shot contract: plx_filter (build for one MVM + that MVM) per call
        wall_cold = time_region(lambda i: lat.filter_once(v, ref, RBF1, out), ncold, ctx.sync, ctx.barrier)
        lat.set_timing(True)
        lat.build(ref, RBF1)
        build_ms = lat.build_times_ms()
        lat.set_timing(False)
        build_ms.pop("csr", None)
        # the splat / slice tables are built by their first user; plx_prepare builds them now, so they can be timed
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        lat.prepare(vd)
        ev1.record()
        ctx.sync()
        build_ms["tables"] = ev0.elapsed_time(ev1)
        kt = kernel_times(lat, v, out, reps=max(10, args.steps), mvm_ms=wall_warm / args.steps * 1e3)
        roof, stages = roofline_for(lat, kt, n_local, d, m, vd, r, args.ell)
        result["warm_mvms_per_s"] = round(args.steps / wall_warm, 1)
        result["cold_mvms_per_s"] = round(ncold / wall_cold, 1)
        result["roofline"] = roof
        result["stages"] = stages
        result["build_ms"] = {k: round(t, 3) for k, t in build_ms.items()}
        result["lattice_device_MB"] = round(lat.device_bytes / 1e6, 1)
        log("stage times (ms):", kt)
        log("build (ms):", build_ms)
        # the same MVM with rows in lattice order, as a CG solve runs it (solvers.khat_solve permutes the right-hand side
        # once per solve, every iteration's MVM skips the two row permutations): reported, never `value`
        lat.set_lattice_row_order(True)
        v_l = lat.to_lattice_order(v)
        for _ in range(3):
            lat.apply(v_l, out)
        wall_l = time_region(lambda i: lat.apply(v_l, out), args.steps, ctx.sync, ctx.barrier)
        kt_l = kernel_times(lat, v_l, out, reps=max(10, args.steps), mvm_ms=wall_l / args.steps * 1e3)
        _, stages_l = roofline_for(lat, kt_l, n_local, d, m, vd, r, args.ell)
        lat.set_lattice_row_order(False)
        result["lattice_row_order"] = {"warm_mvms_per_s": round(args.steps / wall_l, 1),
                                       "stages": {k: {kk: vv for kk, vv in st.items() if kk != "traffic_MB_per_launch"}
                                                  for k, st in stages_l.items()},
                                       "note": "rows of src / out in lattice order (plx_set_row_order): the mode of every "
                                               "CG iteration in solvers.khat_solve; the caller-order figures above "
                                               "include the two row permutations"}
        x_cpu, v_cpu = job.x_cpu, job.v_cpu
        job.close()

        if not args.no_fine:
            # ---- fine regime: same points, lengthscale 0.25 -> m ~ 8.9e6, blur streams from HBM
            ref_f = (x_cpu / 0.25).contiguous().to(ctx.dev)
            lat_f = plx.Lattice(ctx.dev)
            lat_f.build(ref_f, RBF1)                   # (sizes the object's buffers: a first build pays the allocations)
            lat_f.prepare(vd)
            lat_f.set_timing(True)
            lat_f.build(ref_f, RBF1)
            fine_build = lat_f.build_times_ms()
            lat_f.set_timing(False)
            fine_build.pop("csr", None)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            lat_f.prepare(vd)
            ev1.record()
            ctx.sync()
            fine_build["tables"] = ev0.elapsed_time(ev1)
            for _ in range(3):
                lat_f.apply(v, out)
            wf = time_region(lambda i: lat_f.apply(v, out), 20, ctx.sync, ctx.barrier)
            ktf = kernel_times(lat_f, v, out, reps=10, mvm_ms=wf / 20 * 1e3)
            abf = alg_bytes(n_local, d, lat_f.m, vd, r)
            blur_gbps = abf["blur_axis"] / (ktf["blur"] * 1e-3) / 1e9
            _, stages_f = roofline_for(lat_f, ktf, n_local, d, lat_f.m, vd, r, 0.25)
            names_f = lat_f.stage_kernels(vd)
            pmc_f = pmc_traffic(names_f["blur_axis"], 0.25)
            result["fine"] = {
                "lengthscale": 0.25, "m_vertices": lat_f.m, "warm_mvms_per_s": round(20 / wf, 1),
                "blur_roofline": {"bound": "hbm", "kernel": " + ".join(names_f["blur_axis"]),
                                  "achieved": round(blur_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(blur_gbps / HBM_PEAK_GBPS, 4),
                                  "bytes_per_launch": int(abf["blur_axis"]), "launch_us": round(ktf["blur"] * 1e3, 2),
                                  "traffic": pmc_f["bytes"] if pmc_f else None,
                                  "traffic_source": pmc_f["source"] if pmc_f else None,
                                  "achieved_on_traffic": round(pmc_f["bytes"] / (ktf["blur"] * 1e-3) / 1e9, 1) if pmc_f else None,
                                  "frac_on_traffic": round(pmc_f["bytes"] / (ktf["blur"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if pmc_f else None},
                "stages": stages_f, "build_ms": {k: round(t, 3) for k, t in fine_build.items()},
                "lattice_device_MB": round(lat_f.device_bytes / 1e6, 1),
            }
            log("fine stage times (ms):", ktf)
            lat_f.close()
            del ref_f

        if not args.no_configs and args.n == 1_000_000 and d == 8:
            result.update(config3_leg(ctx))
            result["config4"] = sharded_leg(ctx, 4_000_000, d, 1.0, [1, 11], 20)
            result.update(config5_leg(ctx))
            result.update(published_shapes_leg(ctx))

        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(x_cpu[:n_local], v_cpu[:n_local], args.ell)
        if "config3" in result:
            result["config3"]["launches_per_cg_iteration"] = cg_launch_leg(ctx)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
