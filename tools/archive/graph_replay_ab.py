#!/usr/bin/env python3
"""Eager launches against a captured HIP graph (torch.cuda.CUDAGraph) of one MVM, us per MVM, for a few lattice shapes.
    python tools/graph_replay_ab.py"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
import bench

MATERN3 = np.array([0.08435782, 0.24239115, 0.60311586, 1.0, 0.60311586, 0.24239115, 0.08435782], np.float32)
cases = [("config 5 stand-in (N=10623, d=18, Matern order 3)", 10623, 18, 1, 1.0, MATERN3),
         ("config 2 (N=1e5, d=4, l=1)", 100000, 4, 1, 1.0, bench.RBF1),
         ("config 2 (N=1e5, d=4, l=0.25)", 100000, 4, 1, 0.25, bench.RBF1),
         ("headline (N=1e6, d=8, l=1)", 1000000, 8, 1, 1.0, bench.RBF1),
         ("CG iteration (N=1e6, d=8, l=0.6931, 12 columns)", 1000000, 8, 12, 0.6931, bench.RBF1)]
for name, n, d, vd, ell, taps in cases:
    x, v = bench.synth(n, d, vd)
    ref = (x / ell).contiguous().cuda(); v = v.cuda(); out = torch.empty_like(v)
    lat = plx.Lattice().build(ref, taps)
    lat.prepare(vd)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            lat.apply(v, out)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lat.apply(v, out)
    def timeit(fn, reps=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    eager = min(timeit(lambda: lat.apply(v, out)) for _ in range(3))
    replay = min(timeit(graph.replay) for _ in range(3))
    print(json.dumps({"case": name, "m": lat.m, "eager_us": round(eager, 1), "graph_us": round(replay, 1)}), flush=True)
    lat.close()
