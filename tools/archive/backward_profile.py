#!/usr/bin/env python3
"""Stage times of the position-gradient filter (2L(1+d) columns), fused vs three-call form.  N=1e6, d=8, L=11."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx

n, d, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 8, 11
g = torch.Generator().manual_seed(1234)
x = torch.randn(n, d, generator=g)
dk = plx.DiscretizedKernelFN(plx.rbf, 1)
taps = dk.get_deriv_coeffs()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for ell in (1.0, 0.6931):
    ref = (x / ell).contiguous().cuda()
    src = torch.randn(n, L, generator=g).cuda()
    go = torch.randn(n, L, generator=g).cuda()
    lat = plx.Lattice().build(ref, taps)
    lat.set_timing(True)
    for rep in range(3):
        t0 = sync()
        gr, gs = lat.apply_backward(go, src, ref)
        t1 = sync()
    fused = lat.apply_times_ms()
    stacked = torch.cat([go, (go[..., None] * ref[..., None, :]).reshape(n, -1), src, (src[..., None] * ref[..., None, :]).reshape(n, -1)], 1).contiguous()
    for rep in range(3):
        t2 = sync()
        out = lat.apply(stacked)
        t3 = sync()
    unfused = lat.apply_times_ms()
    print(json.dumps({"ell": ell, "m": lat.m, "fused_ms": round((t1 - t0) * 1e3, 3), "fused_stages": fused,
                      "filter_ms": round((t3 - t2) * 1e3, 3), "filter_stages": unfused}), flush=True)
    lat.close()
    del stacked, out
