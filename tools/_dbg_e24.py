import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplex_gp_amd as plx
from simplex_gp_amd import _native as nv
lib = nv.lib()
taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
rng = np.random.default_rng(77)
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for n, d, scale in [(30011, 4, 1.0), (20000, 8, 2.0), (3000, 4, 1.0), (1500, 4, 1.0)]:
    ref = torch.from_numpy((rng.standard_normal((n, d)) / scale).astype(np.float32)).cuda()
    s = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).cuda()
    lib.plx_tune(b"block_path", 0)
    a = plx.Lattice().build(ref, taps)
    va = a.splat(s).clone(); bl = a.blur(va.clone(), vd=1); oa = a.slice(bl, vd=1).clone()
    for e in (16, 24):
        lib.plx_tune(b"block_path", 2); lib.plx_tune(b"block_e", e)
        b = plx.Lattice().build(ref, taps)
        vb = b.splat(s)
        ob = b.slice(bl, vd=1)
        bad = (vb - va).abs() > 1e-3 * va.abs().max()
        print(n, d, "E", e, "rows", b.block_rows, "splat rel", rel(vb, va), "slice rel", rel(ob, oa), "bad vertices", int(bad.sum()), "first bad", bad.nonzero()[:5].flatten().tolist(), flush=True)
        b.close()
    a.close()
