"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the dev container (needs /root/reference and oracle/_ref, see
oracle/build_ref.py).  The outputs are data: inputs + expected outputs.

  filter_small.npz   native filter(src, ref, coeffs) (cpp:6-16 -> h:259-340):
                     inputs, taps, output and vertex count m for ~60 cases
  filter_large.npz   seeds + m + output probes (head / strided samples / norms)
                     for the BASELINE.json shapes (inputs are re-generated from
                     the seed at test time; head of the inputs stored to prove
                     the re-generation matches)
  stages_small.npz   what the reference holds BETWEEN the stages of filter() on 9 of the small cases:
                     vertex keys, values after splat and after blur, per-point greedy / rank
                     (dumped through oracle/ref_stage_driver.cpp, which includes the reference header in place)
  host_side.npz      bilateral_kernel.py: tap vectors (get_coeffs, py:14-28,
                     py:162-181) and LatticeFilterGeneral forward / backward
                     (py:76-124) on small inputs

The vertex count m comes from the -DDEBUG build's "Hash table size" line
(h:300), captured from its stdout.

The reference's Python module imports gpytorch at module scope (py:9-10) and
gpytorch is not installed here, so the two names it binds (Kernel, LazyTensor)
are registered as inert placeholders before the import; none of the functions
captured below touch them.
"""
import io
import os
import re
import sys
import types
import contextlib

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

RBF1 = [0.34608543, 1.0, 0.34608543]
RBF2 = [0.08263808, 0.53616077, 1.0, 0.53616077, 0.08263808]
RBF3 = [0.01831428, 0.16900772, 0.64117509, 1.0, 0.64117509, 0.16900772, 0.01831428]
MAT3 = [0.08435782, 0.24239115, 0.60311586, 1.0, 0.60311586, 0.24239115, 0.08435782]
TAPS = {1: RBF1, 2: RBF2, 3: RBF3}


def capture_m(mod_dbg, src, ref, coeffs):
    """Run the DEBUG build and parse 'Hash table size: m' from its stdout."""
    sys.stdout.flush()
    r, w = os.pipe()
    saved = os.dup(1)
    os.dup2(w, 1)
    try:
        mod_dbg.filter(src, ref, coeffs)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(w)
        os.close(saved)
    text = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        text += chunk
    os.close(r)
    return int(re.search(rb"Hash table size: (\d+)", text).group(1))


def synth(n, d, vd, ell, seed=1234, dist="randn"):
    """SURVEY 8(d) recipe: x first, then v, from one generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g) if dist == "randn" else torch.rand(n, d, generator=g)
    v = torch.randn(n, vd, generator=g)
    return v, (x / ell).contiguous()


def large_record(large, name, n, d, vd, ell, v, r, taps_t, ref, dbg):
    out = ref.filter(v, r, taps_t).numpy()
    m = capture_m(dbg, v, r, taps_t)
    stride = max(1, n // 4096)
    large[f"{name}/shape"] = np.array([n, d, vd], np.int64)
    large[f"{name}/ell"] = np.float64(ell)
    large[f"{name}/seed"] = np.int64(1234)
    large[f"{name}/taps"] = taps_t.numpy()
    large[f"{name}/m"] = np.int64(m)
    large[f"{name}/ref_head"] = r[:8].numpy()
    large[f"{name}/src_head"] = v[:8].numpy()
    large[f"{name}/out_head"] = out[:512]
    large[f"{name}/out_strided"] = out[::stride]
    large[f"{name}/stride"] = np.int64(stride)
    large[f"{name}/out_l2"] = np.float64(np.linalg.norm(out.astype(np.float64)))
    large[f"{name}/out_sum"] = np.float64(out.astype(np.float64).sum())
    large[f"{name}/out_abs_sum"] = np.float64(np.abs(out.astype(np.float64)).sum())
    print(name, "m =", m)


def config_probes():
    """`make_golden.py --config-probes` (round 4): reference probes for BASELINE.json configs[3] and configs[4], ADDED to the
    existing filter_large.npz (the other records are carried over byte for byte):
      config5_n10623_d18_matern3   x, v = synth(10623, 18, 1, 1.0); Matern-1.5 order-3 taps (the stand-in for elevators)
      config4_n4e6_d8_ell1.0       bench.synth(4_000_000, 8, 11): rows in blocks of 1e6, block b from seed 1234 + b
                                   (what every rank of the sharded job generates); column 0 of v"""
    build_ref.build()
    ref = build_ref.load("cpu_lattice_ref")
    dbg = build_ref.load("cpu_lattice_ref_dbg")
    path = os.path.join(HERE, "filter_large.npz")
    old = np.load(path)
    large = {k: old[k] for k in old.files}
    v, r = synth(10623, 18, 1, 1.0)
    large_record(large, "config5_n10623_d18_matern3", 10623, 18, 1, 1.0, v, r, torch.tensor(MAT3, dtype=torch.float32), ref, dbg)
    import bench
    x, v11 = bench.synth(4_000_000, 8, 11)
    large_record(large, "config4_n4e6_d8_ell1.0", 4_000_000, 8, 1, 1.0, v11[:, :1].contiguous(), x.contiguous(),
                 torch.tensor(RBF1, dtype=torch.float32), ref, dbg)
    np.savez_compressed(path, **large)


def main():
    build_ref.build()
    ref = build_ref.load("cpu_lattice_ref")
    dbg = build_ref.load("cpu_lattice_ref_dbg")

    # ------------------------------------------------------------ small cases
    cases = {}

    def add(name, src, refpos, taps):
        taps_t = torch.tensor(taps, dtype=torch.float32)
        out = ref.filter(src, refpos, taps_t)
        m = capture_m(dbg, src, refpos, taps_t)
        cases[name] = dict(src=src.numpy(), ref=refpos.numpy(), taps=taps_t.numpy(),
                           out=out.numpy(), m=np.int64(m))

    # lattice_test.py:9-14 recipe scaled down (src last column = 1)
    g = torch.Generator().manual_seed(0)
    src = torch.randn(2000, 3, generator=g)
    src[:, -1] = 1
    refpos = torch.randn(2000, 6, generator=g)
    add("lattice_test_recipe", src, refpos, [0.5, 1.0, 0.5])

    # experiments/cuda_test.py:10, :62-64 recipe
    g = torch.Generator().manual_seed(1)
    refpos = torch.rand(1000, 10, generator=g)
    src = torch.randn(1000, 1, generator=g)
    add("cuda_test_recipe", src, refpos, [0.5, 1.0, 0.5])

    # Snelson (tests/train_snelson.py data), two lengthscales
    sn = np.loadtxt(os.path.join(HERE, "snelson.csv"), delimiter=",", skiprows=1).astype(np.float32)
    sx, sy = torch.from_numpy(sn[:, :1].copy()), torch.from_numpy(sn[:, 1:].copy())
    for ell in (0.6931, 0.3):
        add(f"snelson_y_ell{ell}", sy, (sx / ell).contiguous(), RBF1)
        eye = torch.eye(200)[:, ::25].contiguous()          # 8 one-hot columns
        add(f"snelson_eye_ell{ell}", eye, (sx / ell).contiguous(), RBF1)

    # seeded clouds: every d x order x ell at N=256, vd=3; a few with N=2000, vd in {1, 11}
    for d in (1, 2, 3, 4, 8, 18):
        for order in (1, 2, 3):
            for ell in (1.0, 0.25):
                v, r = synth(256, d, 3, ell, seed=100 + d)
                add(f"cloud_n256_d{d}_o{order}_ell{ell}", v, r, TAPS[order])
    for d, vd, ell, order in [(2, 1, 1.0, 1), (4, 11, 0.25, 1), (8, 1, 1.0, 1), (8, 3, 0.25, 2),
                              (3, 11, 1.0, 3)]:
        v, r = synth(2000, d, vd, ell, seed=7)
        add(f"cloud_n2000_d{d}_vd{vd}_o{order}_ell{ell}", v, r, TAPS[order])
    # crosses the first hash-table doubling (m > 2^14 - 1): pins the stale-bucket quirk
    v, r = synth(2000, 8, 3, 0.25, seed=1234)
    add("cloud_grow_quirk_n2000_d8", v, r, RBF1)
    # Matern-1.5 order-3 taps, d=18 (config 5 shape, small N)
    v, r = synth(256, 18, 1, 1.0, seed=5)
    add("matern_o3_d18", v, r, MAT3)

    # degenerate inputs
    add("all_identical", torch.randn(64, 2, generator=torch.Generator().manual_seed(3)),
        torch.full((64, 3), 0.37), RBF1)
    add("single_point", torch.tensor([[1.5, -2.0]]), torch.tensor([[0.3, -0.7, 1.1, 0.0]]), RBF1)
    add("origin_ties", torch.ones(16, 1), torch.zeros(16, 4), RBF1)
    gx = torch.arange(-4, 5, dtype=torch.float32)
    grid = torch.stack(torch.meshgrid(gx, gx, indexing="ij"), -1).reshape(-1, 2) * 0.5
    add("grid_ties_d2", torch.ones(grid.shape[0], 2), grid.contiguous(), RBF2)

    flat = {}
    for name, c in cases.items():
        for k, a in c.items():
            flat[f"{name}/{k}"] = a
    np.savez_compressed(os.path.join(HERE, "filter_small.npz"), **flat)
    print("filter_small.npz:", len(cases), "cases")

    # ------------------------------------------------------------ per-stage dumps
    # oracle/ref_stage_driver.cpp drives the reference header's public members (splat, blur, slice,
    # hashTable, greedy, rank) and records what lies between the stages of filter()
    stg = build_ref.load("ref_stages")
    stages = {}
    for name in ["lattice_test_recipe", "cuda_test_recipe", "cloud_n256_d4_o2_ell0.25", "cloud_n256_d8_o1_ell1.0",
                 "cloud_n2000_d3_vd11_o3_ell1.0", "matern_o3_d18", "grid_ties_d2", "origin_ties",
                 "cloud_grow_quirk_n2000_d8"]:
        c = cases[name]
        keys, v_splat, v_blur, out, greedy, rank = stg.stages(torch.from_numpy(c["src"]), torch.from_numpy(c["ref"]),
                                                              torch.from_numpy(c["taps"]))
        assert np.array_equal(out.numpy(), c["out"]), name          # the driver replays filter() exactly
        assert keys.shape[0] == int(c["m"]), name
        stages.update({f"{name}/keys": keys.numpy(), f"{name}/values_after_splat": v_splat.numpy(),
                       f"{name}/values_after_blur": v_blur.numpy(), f"{name}/greedy": greedy.numpy(),
                       f"{name}/rank": rank.numpy()})
    np.savez_compressed(os.path.join(HERE, "stages_small.npz"), **stages)
    print("stages_small.npz:", len(stages) // 5, "cases")

    # ------------------------------------------------------------ large cases
    large = {}
    for name, (n, d, vd, ell) in {
        "n1e5_d4_ell1.0": (100000, 4, 1, 1.0),
        "n1e5_d4_ell0.25": (100000, 4, 1, 0.25),
        "n1e5_d4_vd11_ell1.0": (100000, 4, 11, 1.0),
        "n1e6_d8_ell1.0": (1000000, 8, 1, 1.0),
        "n1e6_d8_ell0.6931": (1000000, 8, 1, 0.6931),
    }.items():
        v, r = synth(n, d, vd, ell)
        taps_t = torch.tensor(RBF1, dtype=torch.float32)
        out = ref.filter(v, r, taps_t).numpy()
        m = capture_m(dbg, v, r, taps_t)
        stride = max(1, n // 4096)
        large[f"{name}/shape"] = np.array([n, d, vd], np.int64)
        large[f"{name}/ell"] = np.float64(ell)
        large[f"{name}/seed"] = np.int64(1234)
        large[f"{name}/taps"] = taps_t.numpy()
        large[f"{name}/m"] = np.int64(m)
        large[f"{name}/ref_head"] = r[:8].numpy()
        large[f"{name}/src_head"] = v[:8].numpy()
        large[f"{name}/out_head"] = out[:512]
        large[f"{name}/out_strided"] = out[::stride]
        large[f"{name}/stride"] = np.int64(stride)
        large[f"{name}/out_l2"] = np.float64(np.linalg.norm(out.astype(np.float64)))
        large[f"{name}/out_sum"] = np.float64(out.astype(np.float64).sum())
        large[f"{name}/out_abs_sum"] = np.float64(np.abs(out.astype(np.float64)).sum())
        print(name, "m =", m)
    np.savez_compressed(os.path.join(HERE, "filter_large.npz"), **large)

    # -------------------------------------------------------------- host side
    class _Kernel(torch.nn.Module):          # placeholder for gpytorch.kernels.Kernel
        def __init__(self, *a, **k):
            super().__init__()

    class _LazyTensor:                       # placeholder for gpytorch.lazy.LazyTensor
        def __init__(self, *a, **k):
            pass

    gp = types.ModuleType("gpytorch")
    gp.kernels = types.ModuleType("gpytorch.kernels")
    gp.kernels.Kernel = _Kernel
    gp.lazy = types.ModuleType("gpytorch.lazy")
    gp.lazy.LazyTensor = _LazyTensor
    sys.modules.update({"gpytorch": gp, "gpytorch.kernels": gp.kernels, "gpytorch.lazy": gp.lazy})
    sys.path.insert(0, "/root/reference")
    with contextlib.redirect_stdout(io.StringIO()):
        import gpytorch_lattice_kernel.bilateral_kernel as bk

    host = {}
    profiles = {
        "rbf": bk.rbf,
        "matern15": lambda d2: bk.Matern.apply(d2, 1.5),
        "matern25": lambda d2: bk.Matern.apply(d2, 2.5),
    }
    dk = {}
    for pname, fn in profiles.items():
        for order in (1, 2, 3):
            with contextlib.redirect_stdout(io.StringIO()):
                k = bk.DiscretizedKernelFN(fn, order)
            dk[(pname, order)] = k
            host[f"coeffs/{pname}_o{order}/fwd"] = k.get_coeffs().detach().numpy()
            host[f"coeffs/{pname}_o{order}/deriv"] = k.get_deriv_coeffs().detach().numpy()

    # LatticeFilterGeneral forward + both gradients through the reference CPU module
    bk.LatticeFilterGeneral.method = ref.filter
    for cname, (n, d, L, pname, order) in {
        "n50_d3_L2_rbf_o1": (50, 3, 2, "rbf", 1),
        "n200_d1_L1_rbf_o1": (200, 1, 1, "rbf", 1),
        "n50_d3_L2_matern15_o3": (50, 3, 2, "matern15", 3),
        "n200_d1_L1_matern15_o3": (200, 1, 1, "matern15", 3),
        "n300_d4_L3_rbf_o2": (300, 4, 3, "rbf", 2),
    }.items():
        g = torch.Generator().manual_seed(42)
        x = torch.randn(n, d, generator=g).requires_grad_(True)
        s = torch.randn(n, L, generator=g).requires_grad_(True)
        gout = torch.randn(n, L, generator=g)
        out = bk.LatticeFilterGeneral.apply(s, x, dk[(pname, order)])
        out.backward(gout)
        host[f"autograd/{cname}/x"] = x.detach().numpy()
        host[f"autograd/{cname}/src"] = s.detach().numpy()
        host[f"autograd/{cname}/grad_out"] = gout.numpy()
        host[f"autograd/{cname}/out"] = out.detach().numpy()
        host[f"autograd/{cname}/grad_src"] = s.grad.numpy()
        host[f"autograd/{cname}/grad_x"] = x.grad.numpy()
        # source-only gradient takes the other branch (py:110-111)
        s2 = s.detach().clone().requires_grad_(True)
        out2 = bk.LatticeFilterGeneral.apply(s2, x.detach(), dk[(pname, order)])
        out2.backward(gout)
        host[f"autograd/{cname}/grad_src_only"] = s2.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "host_side.npz"), **host)
    print("host_side.npz:", len(host), "arrays")

    dataset_fixture()


def dataset_fixture():
    # -------------------------------------------------------------- dataset conventions (experiments/utils.py)
    # The reference's own prepare_dataset / UCIDataset / EarlyStopper, imported with `wandb` stubbed (it is only
    # imported, never called on this path), run on a synthetic .mat: the three standardised splits and an
    # early-stopping trace become the fixture that simplex_gp_amd.data / training.EarlyStopper are compared with.
    import tempfile
    from scipy.io import savemat
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    sys.path.insert(0, "/root/reference/experiments")
    import utils as ref_utils
    rng = np.random.default_rng(20211)
    raw = rng.standard_normal((1237, 7)) * np.array([1, 5, 0.1, 3, 2, 1e-3, 10]) + np.array([0, 1, -2, 3, 0, 7, 5])
    raw[:, 2] = np.round(raw[:, 2], 1)                     # a nearly constant, quantised feature (std ~ 0.1)
    ds = {"raw": raw}
    with tempfile.TemporaryDirectory() as tmp:
        savemat(os.path.join(tmp, "toyset.mat"), {"data": raw})
        for mode, x, y in ref_utils.prepare_dataset("toyset", uci_data_dir=tmp, device="cpu"):
            ds[f"{mode}/x"], ds[f"{mode}/y"] = x.numpy(), y.numpy()
        for mode, x, y in ref_utils.prepare_dataset("toyset", uci_data_dir=tmp, device="cpu", train_val_split=0.6):
            ds[f"split0.6/{mode}/x"], ds[f"split0.6/{mode}/y"] = x.numpy(), y.numpy()
    scores = np.array([-1.0, -0.8, -0.80005, -0.7, -0.75, -0.71, -0.6999, -0.9, -0.5, -0.6, -0.55, -0.51, -0.52], np.float64)
    st = ref_utils.EarlyStopper(patience=3, delta=1e-4)
    best, done = [], []
    for i, sc in enumerate(scores):
        if st.is_done():
            break
        st(float(sc), i)
        best.append(st.info())
        done.append(st.is_done())
    ds["stopper/scores"], ds["stopper/best"], ds["stopper/done"] = scores, np.array(best), np.array(done)
    np.savez_compressed(os.path.join(HERE, "dataset_split.npz"), **ds)
    print("dataset_split.npz:", sorted(ds))


if __name__ == "__main__" and "--config-probes" in sys.argv:
    config_probes()
    sys.exit(0)

if __name__ == "__main__":
    # `python tests/golden/make_golden.py dataset` regenerates dataset_split.npz only
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        dataset_fixture()
    else:
        main()
