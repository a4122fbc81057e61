#!/usr/bin/env python3
"""Checker for `bench.py --dump PREFIX` (multi-rank rehearsals): regenerates the bench's synthetic inputs, runs the
CPU oracle on the whole problem and compares every rank's dumped output rows.  Test infrastructure: lives under
tests/ because bench.py may touch oracle/ only for its cpu_baseline.

    python -m torch.distributed.run --nproc-per-node 3 ... bench.py --gpus 3 --points 60000 --backend gloo --dump gpurun_out/d
    python tests/check_bench_dump.py gpurun_out/d
"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                      # noqa: E402  (synth(): the same seeded inputs)
from oracle import oracle        # noqa: E402


def main():
    files = sorted(glob.glob(sys.argv[1] + ".rank*.npz"))
    assert files, "no dumps found"
    z0 = np.load(files[0])
    n_total, d, vd, ell = int(z0["n_total"]), int(z0["d"]), int(z0["vd"]), float(z0["ell"])
    x, v = bench.synth(n_total, d, vd)
    oracle.set_exact_mode(False)
    want = oracle.filter(v.numpy(), (x / ell).numpy(), bench.RBF1)
    worst = 0.0
    for f in files:
        z = np.load(f)
        lo, hi = int(z["lo"]), int(z["hi"])
        err = float(np.linalg.norm(z["out"].astype(np.float64) - want[lo:hi]) / np.linalg.norm(want[lo:hi]))
        worst = max(worst, err)
        print(f"{os.path.basename(f)}: rows [{lo},{hi}) rel-L2 vs oracle {err:.2e} (m={int(z['m'])})")
    assert worst <= 1e-5, worst
    print("OK")


if __name__ == "__main__":
    main()
