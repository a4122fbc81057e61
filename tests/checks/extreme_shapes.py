import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import simplex_gp_amd as plx
from oracle import oracle
oracle.set_exact_mode(False)
rng = np.random.default_rng(1)
worst = 0
for (n, d, vd, order) in [(500, 32, 1, 1), (500, 32, 5, 2), (2000, 25, 3, 3), (300, 31, 130, 1), (1000, 3, 1, 4), (1000, 3, 7, 5), (1000, 2, 40, 6), (1000, 4, 1, 8), (700, 5, 130, 7), (64, 1, 1, 8)]:
    ref = (rng.standard_normal((n, d)) * 0.8).astype(np.float32)
    src = rng.standard_normal((n, vd)).astype(np.float32)
    half = np.exp(-0.5 * (np.arange(order, 0, -1) * 0.7) ** 2).astype(np.float32)
    taps = np.concatenate([half, [1.0], half[::-1]]).astype(np.float32)
    try:
        got = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), torch.from_numpy(taps)).cpu().numpy()
        # the many-MVM build with the Morton vertex numbering forced (2 code bits per coordinate at d = 32)
        from simplex_gp_amd import _native as nv
        nv.check(nv.lib().plx_tune(b"vertex_order", 2), "plx_tune")
        lat = plx.Lattice().build(torch.from_numpy(ref).cuda(), taps)
        got2 = lat.apply(torch.from_numpy(src).cuda()).cpu().numpy()
        assert lat.stage_kernels()["vertex_order"] == ["morton"] or lat.m < 2
        lat.close()
        nv.check(nv.lib().plx_tune(b"vertex_order", 1), "plx_tune")
        assert np.linalg.norm(got2.astype(np.float64) - got) <= 2e-6 * np.linalg.norm(got), "Morton numbering changed the result"
    except Exception as e:
        print((n, d, vd, order), "HIP raised:", type(e).__name__, str(e)[:120]); continue
    try:
        want = oracle.filter(src, ref, taps)
    except Exception as e:
        print((n, d, vd, order), "oracle raised:", type(e).__name__, str(e)[:120]); continue
    err = float(np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want))
    worst = max(worst, err)
    print((n, d, vd, order), f"rel-L2 {err:.2e}", flush=True)
print("worst", worst)
