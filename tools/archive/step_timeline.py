#!/usr/bin/env python3
"""Where does a rebuild step of bench.py's loop spend its time?  Host timestamps around the build call, the first MVM
after it and the following ones, with and without a device synchronisation between them.
    python tools/step_timeline.py [--n 1000000] [--ell 1.0]"""
import argparse, os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--ell", type=float, default=1.0)
args = ap.parse_args()
x, v = bench.synth(args.n, 8, 1)
ref = (x / args.ell).contiguous().cuda(); v = v.cuda(); out = torch.empty_like(v)
lat = plx.Lattice()
sync = torch.cuda.synchronize
now = time.perf_counter
for _ in range(3):
    lat.build(ref, bench.RBF1)
    for _ in range(60):
        lat.apply(v, out)
sync()
rows = []
for trial in range(6):
    for _ in range(50):                       # a queue of MVMs in flight, as in the bench loop
        lat.apply(v, out)
    t0 = now(); lat.build(ref, bench.RBF1); t1 = now()      # returns when the last read-back has arrived
    sync(); t2 = now()
    lat.apply(v, out); t3 = now(); sync(); t4 = now()       # first MVM: builds its tables
    lat.apply(v, out); sync(); t5 = now()
    lat.apply(v, out); sync(); t6 = now()
    for _ in range(50):
        lat.apply(v, out)
    sync(); t7 = now()
    rows.append({"build_call_us": (t1 - t0) * 1e6, "build_tail_us": (t2 - t1) * 1e6, "first_mvm_call_us": (t3 - t2) * 1e6,
                 "first_mvm_us": (t4 - t2) * 1e6, "second_mvm_us": (t5 - t4) * 1e6, "third_mvm_us": (t6 - t5) * 1e6,
                 "next50_us_each": (t7 - t6) / 50 * 1e6})
for r in rows:
    print(json.dumps({k: round(v, 1) for k, v in r.items()}), flush=True)
# the bench loop itself, no synchronisation inside
def loop(k):
    for i in range(k):
        if i % 50 == 0:
            lat.build(ref, bench.RBF1)
        lat.apply(v, out)
for _ in range(2):
    loop(100)
sync(); t0 = now(); loop(100); sync(); t1 = now()
print(json.dumps({"loop_us_per_step": round((t1 - t0) / 100 * 1e6, 1)}))
# and with the build's 50-MVM queue drained first (what the build costs with an idle GPU)
sync(); t0 = now()
for _ in range(10):
    lat.build(ref, bench.RBF1); sync()
t1 = now()
print(json.dumps({"idle_gpu_build_us": round((t1 - t0) / 10 * 1e6, 1)}))
