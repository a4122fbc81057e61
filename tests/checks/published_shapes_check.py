#!/usr/bin/env python3
"""The largest shape of the reference's published timing table (houseelectric: n = 2,049,280, d = 11) at FULL size against
the CPU oracle -- three minutes of oracle time on one host core, so a check script rather than a test.
    python tests/checks/published_shapes_check.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from oracle import oracle
import bench

n, d = 2_049_280, 11
g = torch.Generator().manual_seed(1234)
ref = (torch.randn(n, d, generator=g) / 0.6931).contiguous()
v = torch.randn(n, 1, generator=g)
lat = plx.Lattice().build(ref.cuda(), bench.RBF1)
out = lat.apply(v.cuda()).cpu().numpy()
t0 = time.time()
oracle.set_exact_mode(False)
want, m = oracle.filter(v.numpy(), ref.numpy(), bench.RBF1, return_m=True)
oracle.set_exact_mode(True)
err = float(np.linalg.norm(out.astype(np.float64) - want) / np.linalg.norm(want.astype(np.float64)))
print(json.dumps({"n": n, "d": d, "m_hip": lat.m, "m_oracle": int(m), "rel_l2": err, "oracle_s": round(time.time() - t0, 1)}))
assert lat.m == m and err <= 1e-5
