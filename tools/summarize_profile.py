#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/run_profiles.sh) into one markdown file + one JSON table.

Per kernel and grid size: calls, mean / min duration from --kernel-trace of the full bench.py run, and, per
lattice regime (lengthscale 1.0 = bench headline, 0.25 = fine), the per-launch FETCH_SIZE / WRITE_SIZE from the
--pmc passes.  FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM
section); both the raw and the doubled figure are printed (narrow 4-byte gathers are uncalibrated, so for
gather-heavy kernels the truth lies between the two).
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("void ", "")
    base = name.split("(")[0]
    if "rocprim" in base:
        return "rocprim::" + ("onesweep" if "onesweep" in name else base.split("::")[-1][:30])
    return base[:48]


def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def grid_of(r):
    return int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    trace = load(os.path.join(src, "trace", "**", "*kernel_trace.csv"))
    dur = collections.defaultdict(list)
    for r in trace:
        dur[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = ["# rocprofv3 summary (%s)" % os.path.basename(src.rstrip("/")), "",
             "Trace: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-configs`",
             "(both lattice regimes in one process).  Durations in us.", "",
             "| kernel | grid (threads) | calls | mean us | min us |", "|---|---:|---:|---:|---:|"]
    for key, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| %s | %d | %d | %.2f | %.2f |" % (key[0], key[1], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
    table = {}
    for ell in ("1.0", "0.25"):
        pm = {}
        for which in ("FETCH_SIZE", "WRITE_SIZE"):
            acc = collections.defaultdict(list)
            for r in load(os.path.join(src, f"pmc_{ell}_{which}", "**", "*counter_collection.csv")):
                acc[(short(r["Kernel_Name"]), grid_of(r))].append(float(r["Counter_Value"]))
            pm[which] = acc
        lines += ["", f"## PMC per launch, lengthscale {ell} (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, "
                      "`bench.py ... --skip-fine --ell %s`)" % ell, "",
                  "Counter unit KB.  `fetch x2` applies the gfx950 wide-read correction; traffic = fetch x2 + write.", "",
                  "| kernel | grid | fetch MB | fetch x2 MB | write MB | traffic MB |", "|---|---:|---:|---:|---:|---:|"]
        sec = {}
        for key in sorted(set(pm["FETCH_SIZE"]) & set(pm["WRITE_SIZE"]), key=lambda k: -sum(pm["FETCH_SIZE"][k])):
            f, w = pm["FETCH_SIZE"][key], pm["WRITE_SIZE"][key]
            fm, wm = sum(f) / len(f) / 1024, sum(w) / len(w) / 1024
            if fm + wm < 0.05:
                continue
            lines.append("| %s | %d | %.1f | %.1f | %.1f | %.1f |" % (key[0], key[1], fm, 2 * fm, wm, 2 * fm + wm))
            sec[f"{key[0]}|{key[1]}"] = {"fetch_KB": sum(f) / len(f), "write_KB": sum(w) / len(w)}
        table[ell] = sec
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        lines += ["", "## rocprofv3 --stats (kernel_stats.csv, top rows)", "", "```"]
        for i, row in enumerate(csv.reader(open(stats[0]))):
            if i > 14:
                break
            row[0] = short(row[0]) if i else row[0]
            lines.append(",".join(row))
        lines.append("```")
    bench = os.path.join(src, "trace.json")
    if os.path.exists(bench):
        lines += ["", "## bench.py line under the profiler (slower than an un-profiled run)", "", "```", open(bench).read().strip(), "```"]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write("\n".join(lines) + "\n")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per launch, by lattice lengthscale; FETCH_SIZE counts "
                       "64 B per 128-B request on gfx950 wide reads (MI355X_MICROARCH.md, HBM): traffic = 2*fetch + write",
               "kernel_sources_sha16": bench.kernel_sources_sha16(),
               "by_lengthscale": table}, open(dst.replace("_summary.md", "_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
