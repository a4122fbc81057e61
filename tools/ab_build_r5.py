#!/usr/bin/env python3
"""Round 5: lattice-build A/B (N = 1e6, d = 8 unless --n): plx_tune variants interleaved in one process, build-stage times
(minimum over rounds), table-build time (plx_prepare), one MVM's time, and the built structure compared with the first
variant's bit for bit (vertex keys, per-corner vertex ids, neighbour table, output of one MVM).

    python tools/ab_build_r5.py --ells 1.0 0.25 --variants "nbr_sliced=0" "nbr_sliced=1" "nbr_sliced=2"
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--ells", type=float, nargs="+", default=[1.0, 0.6931, 0.25])
ap.add_argument("--variants", nargs="+", default=["nbr_sliced=0", "nbr_sliced=1", "nbr_sliced=2"])
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--no-compare", action="store_true")
args = ap.parse_args()

variants = [dict((k, int(v)) for k, v in (kv.split("=") for kv in spec.split(","))) for spec in args.variants]
all_keys = sorted({k for v in variants for k in v})
defaults = {k: nv.lib().plx_tune_get(k.encode()) if hasattr(nv.lib(), "plx_tune_get") else None for k in all_keys}


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


x, v = bench.synth(args.n, args.d, 1)
vc = v.cuda()
out = torch.empty_like(vc)
for ell in args.ells:
    ref = (x / ell).contiguous().cuda()
    res = [dict() for _ in variants]
    base = None
    for rnd in range(args.rounds):
        for vi, tunes in enumerate(variants):
            for k, val in tunes.items():
                nv.check(nv.lib().plx_tune(k.encode(), val), f"tune {k}")
            r = res[vi]
            lat = r.get("lat") or plx.Lattice()
            r["lat"] = lat
            lat.set_timing(True)
            t0 = sync(); lat.build(ref, bench.RBF1); t1 = sync()
            bt = lat.build_times_ms(); bt.pop("csr", None)
            lat.set_timing(False)
            t2 = sync(); lat.prepare(1); t3 = sync()
            for _ in range(3):
                lat.apply(vc, out)
            t4 = sync()
            for _ in range(20):
                lat.apply(vc, out)
            t5 = sync()
            # the bench's own cadence: one build + 20 MVMs, no synchronisation inside
            t6 = sync(); lat.build(ref, bench.RBF1)
            for _ in range(20):
                lat.apply(vc, out)
            t7 = sync()
            r["wall_build_ms"] = min(r.get("wall_build_ms", 1e9), (t1 - t0) * 1e3)
            r["tables_ms"] = min(r.get("tables_ms", 1e9), (t3 - t2) * 1e3)
            r["mvm_us"] = min(r.get("mvm_us", 1e9), (t5 - t4) / 20 * 1e6)
            r["step20_ms"] = min(r.get("step20_ms", 1e9), (t7 - t6) * 1e3)
            st = r.setdefault("stages", {})
            for k, t in bt.items():
                st[k] = min(st.get(k, 1e9), t)
            r["m"] = lat.m
            if rnd == 0 and not args.no_compare:
                cur = {"keys": lat.export(nv.ARRAY_KEYS), "evid": lat.export(nv.ARRAY_ENTRY_VERTEX),
                       "nbr": lat.export(nv.ARRAY_NEIGHBORS), "out": out.clone().cpu().numpy()}
                if base is None:
                    base = cur
                    r["identical"] = True
                else:
                    r["identical"] = {k: bool(np.array_equal(cur[k], base[k])) for k in cur}
                    r["out_rel_diff"] = float(np.linalg.norm(cur["out"].astype(np.float64) - base["out"]) / np.linalg.norm(base["out"]))
                    r["same_key_set"] = bool(np.array_equal(np.unique(cur["keys"], axis=0), np.unique(base["keys"], axis=0))) if cur["keys"].shape[0] < 2_000_000 else None
                del cur
    for vi, r in enumerate(res):
        r.pop("lat").close()
        st = {k: round(t, 3) for k, t in r.pop("stages").items()}
        print(json.dumps({"ell": ell, "variant": args.variants[vi], "m": r["m"], "stages_ms": st, "stages_total": round(sum(st.values()), 3),
                          **{k: (round(val, 3) if isinstance(val, float) else val) for k, val in r.items() if k != "m"}}), flush=True)
    del ref, base
