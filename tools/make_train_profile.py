#!/usr/bin/env python3
"""profiles/<tag>_train.md: rocprofv3 kernel table of marginal-likelihood training steps (tools/train_step_r4.py) with the
SURVEY 8(d) roofline fractions of the kernels that only the training step runs (the backward pass' wide filter, the
preconditioner's passes).
    tools/make_train_profile.py <tag> <rocprof dir pre_size 0> <rocprof dir pre_size 100> <log 0> <log 100>"""
import csv, glob, json, os, sys

tag, dir0, dir100, log0, log100 = sys.argv[1:6]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, d, L, r, kp = 1_000_000, 8, 11, 1, 112
vd = 2 * L * (1 + d)


def stats(path):
    f = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[-1]
    return list(csv.DictReader(open(f)))


def last_row(log):
    rows = [json.loads(l) for l in open(log) if l.startswith("{")]
    return rows[-1], rows


def alg_bytes(name, m_fwd, m_bwd):
    """algorithmic bytes per launch (SURVEY 8d formulas; fp32 values, int32 ids), or None"""
    if "blur_axis_multi_kernel" in name:
        return m_bwd * (8 * vd + 8 * r), f"m (8 vd + 8 r), vd = {vd}, m = {m_bwd}"
    if "splat_wide_kernel" in name and "StackSource" in name:
        return 4 * N * vd + 8 * N * (d + 1) + 4 * m_bwd * vd, "B_splat at vd = 198 (the stack is formed from packed records, never stored)"
    if "slice_contract" in name:
        return 8 * N * (d + 1) + 4 * m_bwd * vd + 4 * N * (d + L), "B_slice at vd = 198 without the N x vd output, + grad_x, grad_src"
    if "backward_pack_kernel" in name:
        return 4 * N * (2 * L + d) + 4 * N * 32, "g, src, x in; one 128-byte record per point out"
    if "pcg_gram_kernel" in name:
        return 2 * kp * N + 4 * N * 12, "fp16 factor [112][N] + R [N][12]"
    if "pcg_apply_kernel<12, false, true>" in name:
        return 2 * 100 * N + 2 * 4 * N * 12, "fp16 factor [100][N] + R in + Z out"
    if "cg_step_update" in name:
        return 6 * 4 * N * 12, "X, R, P, AP in; X, R out"
    if "step_direction4_kernel" in name or "cg_step_direction_fused" in name:
        return 3 * 4 * N * 12, "P, R (or Z) in; P out"
    if "pchol_multi_step_kernel" in name:
        return 4 * N * (2 * 16 + 3), "16 panel rows in, 16 factor columns out, diagonal in / out, tie-break ranks in"
    if "pcg_apply_kernel<16, true, false>" in name:
        return 4 * N * (16 + 16 + 50), "kernel rows [N][16] in, panel [16][N] out, on average 50 finished factor columns in"
    if "onehot_slice_kernel" in name:
        return 8 * N * (d + 1) + 4 * N * 16, "vertex ids + weights of every corner in, kernel rows [N][16] out (the vertex rows through the position map: mostly absent)"
    return None


def table(rows, m_fwd, m_bwd, steps, top=34):
    out = []
    total = sum(float(x["TotalDurationNs"]) for x in rows)
    for x in rows[:top]:
        name = x["Name"].replace("void ", "").split("(")[0][:72]
        mean = float(x["AverageNs"]) / 1e3
        ab = alg_bytes(x["Name"], m_fwd, m_bwd)
        frac = ""
        if ab:
            gbps = ab[0] / (mean * 1e-6) / 1e9
            frac = f"  {ab[0] / 1e6:8.1f} MB  {gbps:7.0f} GB/s  {gbps / 8000:5.2f} of 8 TB/s   [{ab[1]}]"
        out.append(f"{name:72s} calls/step {int(x['Calls']) / steps:7.1f}  mean {mean:9.2f} us  per step {float(x['TotalDurationNs']) / steps / 1e6:7.3f} ms{frac}")
    launches = sum(int(x["Calls"]) for x in rows) / steps
    out.append(f"(kernel time per step: {total / steps / 1e6:.2f} ms; launches per step: {launches:.0f} -- all {len(rows)} kernel names of the "
               "trace, the build of the step's lattice included)")
    return "\n".join(out)


r0, rows0 = last_row(log0)
r100, rows100 = last_row(log100)


def two(ms):
    """(lattice of the forward / solve, lattice of the backward filter): the RBF step runs both on ONE lattice, and since round 6
    the cache rebuilds that lattice in place when the lengthscale moves, so the log names one."""
    ms = list(ms or [0])
    return (ms + ms)[:2] if len(ms) == 1 else ms[-2:]
prof0 = [x for x in rows0 if "phases_ms" not in x]          # the steps that ran under the profiler (the phase runs are appended to the log)
prof100 = [x for x in rows100 if "phases_ms" not in x]
md = f"""# rocprofv3 of the training step ({tag})

`rocprofv3 --kernel-trace --stats -- python3 tools/train_step_r4.py --pre P --steps S --no-profile`: N = 1e6, d = 8, RBFLattice
order 1, GPyTorch's default initial hyper-parameters, 10 probes, cg_tolerance 1, max 500 CG iterations
(experiments/train_simplexgp.py:29-57); one step = marginal likelihood forward (preconditioner, probes, batched CG with
Lanczos coefficients, SLQ, differentiable MVM) + backward (ONE filter of 2 L (1 + d) = {vd} columns with the derivative
taps, py:113-123; for the RBF profile these are the forward taps, so it runs on the step's one lattice) + Adam.  Adam moves the lengthscale by 10 % per step, so the steps of one run see
different lattices; the table is the mean over the run's steps.  Fractions: SURVEY 8(d) algorithmic bytes / mean launch
time / 8 TB/s.

## pre_size 0 ({len(prof0)} steps under the profiler; lattice(s) in the cache after the last step m = {r0.get('lattices_m')})

```
{table(stats(dir0), *two(r0.get('lattices_m')), len(prof0))}
```

## pre_size 100 ({len(prof100)} steps under the profiler; m = {r100.get('lattices_m')})

```
{table(stats(dir100), *two(r100.get('lattices_m')), len(prof100))}
```

Wall time per step (no profiler; phases from `--steps 3` with the phase synchronisation on):

```
{chr(10).join(json.dumps(x) for x in [y for y in rows0 if 'phases_ms' in y][-2:] + [y for y in rows100 if 'phases_ms' in y][-2:])}
```
"""
open(os.path.join(root, "profiles", f"{tag}_train.md"), "w").write(md)
print("wrote", f"profiles/{tag}_train.md")
