#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/run_profiles.sh) into one markdown file.

Per kernel and grid size: calls, mean / min duration from --kernel-trace, and
the per-launch FETCH_SIZE / WRITE_SIZE from the two --pmc passes.  FETCH_SIZE on
gfx950 reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md,
HBM section); both the raw and the doubled figure are printed and the caveat is
repeated next to them (narrow 4-byte gathers are uncalibrated).
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void ", "")
    base = name.split("(")[0]
    if "rocprim" in base:
        return "rocprim::" + ("onesweep" if "onesweep" in name else base.split("::")[-1][:30])
    return base[:44]


def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    src, dst = sys.argv[1], sys.argv[2]
    trace = load(os.path.join(src, "trace", "**", "*kernel_trace.csv"))
    dur = collections.defaultdict(list)
    for r in trace:
        dur[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = {}
    for which in ("fetch", "write"):
        rows = load(os.path.join(src, which, "**", "*counter_collection.csv"))
        acc = collections.defaultdict(list)
        for r in rows:
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]))
            acc[key].append(float(r["Counter_Value"]))
        pmc[which] = acc
    lines = ["# rocprofv3 summary (%s)" % os.path.basename(src.rstrip("/")), "",
             "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --skip-cpu-baseline`",
             "plus two counter passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with `--kernel-trace` only).",
             "Durations in us from the kernel trace; FETCH/WRITE in MB per launch (counter unit: KB).",
             "gfx950 caveat: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> `fetch x2` column;",
             "4-byte gathers are uncalibrated, so for gather-heavy kernels the truth lies between the two columns.", "",
             "| kernel | grid | calls | mean us | min us | fetch MB | fetch x2 MB | write MB |",
             "|---|---:|---:|---:|---:|---:|---:|---:|"]
    for key, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        f = pmc["fetch"].get(key)
        w = pmc["write"].get(key)
        fm = sum(f) / len(f) / 1024 if f else None
        wm = sum(w) / len(w) / 1024 if w else None
        lines.append("| %s | %d | %d | %.2f | %.2f | %s | %s | %s |" % (
            key[0], key[1], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3,
            "%.1f" % fm if fm is not None else "-", "%.1f" % (2 * fm) if fm is not None else "-",
            "%.1f" % wm if wm is not None else "-"))
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


if __name__ == "__main__":
    main()
