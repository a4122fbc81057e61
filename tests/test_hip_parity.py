"""HIP path vs the CPU oracle and the reference goldens (needs an MI355X).

Everything goes through the C ABI (simplex_gp_amd._native -> libplx.so).
Bars:
  * structure (keys, vertex ids, weights, neighbour table, m): bit-exact against
    the duplicate-free oracle (oracle exact_mode=False; the reference's
    stale-bucket quirk, h:105 vs h:61-63, is documented in DESIGN.md);
  * floating point: rel-L2 <= 1e-5 against the oracle stage by stage, and
    <= 1e-4 against the reference's own output (north_star tolerance) wherever
    the reference quirk itself stays below that.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)

TOL_ORACLE = 1e-5      # HIP vs duplicate-free oracle, rel-L2
TOL_REFERENCE = 1e-4   # HIP vs reference output, rel-L2 (BASELINE.json north_star)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


@pytest.fixture(scope="module")
def plx():
    import simplex_gp_amd as plx
    assert torch.cuda.is_available()
    return plx


@pytest.fixture(scope="module")
def small(golden_dir):
    z = np.load(os.path.join(golden_dir, "filter_small.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


def _clean_oracle(ref, taps):
    oracle.set_exact_mode(False)
    try:
        return oracle.Lattice(ref, taps)
    finally:
        oracle.set_exact_mode(True)


def _relabel(lat, o):
    """Map HIP vertex ids (first touch in lattice point order) to oracle ids (first
    touch in the caller's order) through the vertex keys, which must be the same set."""
    from simplex_gp_amd import _native as nv
    hip_keys = lat.export(nv.ARRAY_KEYS)
    okeys = o.keys
    assert hip_keys.shape == okeys.shape
    lookup = {k.tobytes(): i for i, k in enumerate(okeys)}
    assert len(lookup) == len(okeys)                       # oracle keys are unique
    to_oracle = np.array([lookup[k.tobytes()] for k in hip_keys], np.int64)
    assert len(set(to_oracle.tolist())) == len(to_oracle)  # bijection: no duplicate vertices
    return to_oracle


ROUND5_BUILD_DEFAULTS = {"nbr_sliced": 1, "nbr_seed": 1, "assign_evid": 1, "vertex_order": 1, "nbr_bitmap": 1, "insert_dedupe": 2}


@pytest.mark.parametrize("tunes", [
    {"nbr_sliced": 2},                                         # sliced lookups + neighbours seeded from the embedding
    {"nbr_sliced": 2, "nbr_seed": 0},                          # sliced lookups alone
    {"nbr_sliced": 2, "vertex_order": 2},                      # Morton-numbered: sliced lookups, no seeding
    {"assign_evid": 0, "insert_dedupe": 0},                    # ids through the table lookups, every lane probes
    {"nbr_sliced": 0, "nbr_bitmap": 2},                        # the round-4 lookups (slot bitmap + hashed probes)
    {"assign_evid": 1, "vertex_order": 0},                     # first-touch numbering, ids stored by the numbering pass
], ids=lambda t: ",".join(f"{k}={v}" for k, v in t.items()))
def test_structure_bit_exact_round5_build_paths(plx, small, tunes):
    """Every build variant that is still a switch (XCD-sliced neighbour lookups with and without neighbours seeded from the
    embedding, under either vertex numbering; ids stored by the numbering pass or looked up; the round-4 lookups) builds
    the oracle's structure bit for bit: vertex keys, per-corner vertex ids, the whole neighbour table.  (Round 6 removed
    the measured-loser variants -- the 64-bit mix hash, occupancy-only map nibbles, table-gather flags as a choice, the
    corner-per-thread insert -- together with their code.)"""
    z, names = small
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    try:
        for k, v in tunes.items():
            nv.check(lib.plx_tune(k.encode(), v), "plx_tune")
        lat = plx.Lattice()
        for name in names:
            ref, taps = z[f"{name}/ref"], z[f"{name}/taps"]
            o = _clean_oracle(ref, taps)
            lat.build(torch.from_numpy(ref).cuda(), taps)
            assert lat.m == o.m, name
            to_oracle = _relabel(lat, o)
            perm = lat.export(nv.ARRAY_POINT_PERM).astype(np.int64)
            ev = lat.export(nv.ARRAY_ENTRY_VERTEX)
            assert np.array_equal(to_oracle[ev], o.entry_vertex[perm].T), name
            nbr = lat.export(nv.ARRAY_NEIGHBORS)
            mapped = np.where(nbr >= 0, to_oracle[np.maximum(nbr, 0)], -1)
            assert np.array_equal(mapped, o.neighbors()[:, :, to_oracle]), name
            o.close()
        lat.close()
    finally:
        for k, v in ROUND5_BUILD_DEFAULTS.items():
            nv.check(lib.plx_tune(k.encode(), v), "plx_tune")


def test_structure_bit_exact(plx, small):
    """Keys, per-point simplex corners, barycentric weights, neighbour table and CSR
    against the duplicate-free oracle, bit for bit, modulo the (checked) relabelling
    of vertices and the (exported) reordering of points."""
    z, names = small
    from simplex_gp_amd import _native as nv
    lat = plx.Lattice()
    for name in names:
        ref, taps = z[f"{name}/ref"], z[f"{name}/taps"]
        o = _clean_oracle(ref, taps)
        lat.build(torch.from_numpy(ref).cuda(), taps)
        assert lat.m == o.m, name
        to_oracle = _relabel(lat, o)
        perm = lat.export(nv.ARRAY_POINT_PERM).astype(np.int64)
        assert np.array_equal(np.sort(perm), np.arange(lat.n)), name
        ev = lat.export(nv.ARRAY_ENTRY_VERTEX)             # [d+1, n] in lattice order
        assert np.array_equal(to_oracle[ev], o.entry_vertex[perm].T), name
        assert np.array_equal(lat.export(nv.ARRAY_ENTRY_WEIGHT), o.entry_weight[perm].T), name
        nbr = lat.export(nv.ARRAY_NEIGHBORS)               # [d+1, 2r, m] hip ids
        onbr = o.neighbors()
        mapped = np.where(nbr >= 0, to_oracle[np.maximum(nbr, 0)], -1)
        assert np.array_equal(mapped, onbr[:, :, to_oracle]), name
        # CSR: row lengths are the vertex degrees; every corner appears once
        row_ptr = lat.export(nv.ARRAY_ROW_PTR)
        counts = np.bincount(o.entry_vertex.reshape(-1), minlength=o.m)
        assert np.array_equal(np.diff(row_ptr), counts[to_oracle]), name
        csr_pt = lat.export(nv.ARRAY_CSR_POINT)
        assert np.array_equal(np.bincount(csr_pt, minlength=lat.n), np.full(lat.n, lat.d + 1)), name
        o.close()
    lat.close()


def test_stages_match_oracle(plx, small):
    z, names = small
    from simplex_gp_amd import _native as nv
    lat = plx.Lattice()
    for name in names:
        ref, taps, src = z[f"{name}/ref"], z[f"{name}/taps"], z[f"{name}/src"]
        vd = src.shape[1]
        o = _clean_oracle(ref, taps)
        lat.build(torch.from_numpy(ref).cuda(), taps)
        to_oracle = _relabel(lat, o)
        stride = lat.values_stride(vd)

        def to_hip(ovals):          # oracle-numbered [m, vd] -> hip-numbered, padded [m, stride]
            buf = np.zeros((lat.m, stride), np.float32)
            buf[:, :vd] = ovals[to_oracle]
            return torch.from_numpy(buf).cuda()

        def from_hip(t):            # back to oracle numbering, padding stripped
            out = np.empty((lat.m, vd), np.float32)
            out[to_oracle] = t.cpu().numpy()[:, :vd]
            return out

        s = torch.from_numpy(src).cuda()
        v0 = lat.splat(s)
        assert v0.shape == (lat.m, stride)
        assert float(v0[:, vd:].abs().sum()) == 0.0
        o_v0 = o.splat(src)
        assert rel_l2(from_hip(v0), o_v0) <= TOL_ORACLE, name
        v1 = lat.blur(to_hip(o_v0), vd=vd)
        o_v1 = o.blur(o_v0)
        assert rel_l2(from_hip(v1), o_v1) <= TOL_ORACLE, name
        out = lat.slice(to_hip(o_v1), vd=vd)
        assert out.shape == (lat.n, vd)
        assert rel_l2(out.cpu().numpy(), o.slice(o_v1)) <= TOL_ORACLE, name
        o.close()
    lat.close()


def test_point_order_is_an_implementation_detail(plx):
    """Same output rows with and without the internal spatial sort (plx_tune sort_points)."""
    from simplex_gp_amd import _native as nv
    g = torch.Generator().manual_seed(8)
    x = torch.randn(30000, 5, generator=g).cuda()
    v = torch.randn(30000, 3, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    try:
        nv.check(nv.lib().plx_tune(b"sort_points", 0), "plx_tune")
        a = plx.Lattice().build(x, taps)
        perm = a.export(nv.ARRAY_POINT_PERM)
        assert np.array_equal(perm, np.arange(30000))
        out_a = a.apply(v).clone()
    finally:
        nv.check(nv.lib().plx_tune(b"sort_points", 1), "plx_tune")
    b = plx.Lattice().build(x, taps)
    assert not np.array_equal(b.export(nv.ARRAY_POINT_PERM), np.arange(30000))
    assert b.m == a.m
    assert rel_l2(b.apply(v).cpu().numpy(), out_a.cpu().numpy()) <= 1e-6


def test_filter_vs_reference_goldens(plx, small):
    """End to end through the reference boundary filter(src, ref, coeffs)."""
    z, names = small
    worst = 0.0
    for name in names:
        ref, taps, src, gold = (z[f"{name}/{k}"] for k in ("ref", "taps", "src", "out"))
        out = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(),
                         torch.from_numpy(taps)).cpu().numpy()
        oracle.set_exact_mode(False)
        clean = oracle.filter(src, ref, taps)
        oracle.set_exact_mode(True)
        assert rel_l2(out, clean) <= TOL_ORACLE, name
        quirk = rel_l2(clean, gold)          # what the reference's hash-growth quirk alone costs
        err = rel_l2(out, gold)
        worst = max(worst, err if quirk <= 2e-5 else 0.0)
        assert err <= max(TOL_REFERENCE, quirk + TOL_ORACLE), (name, err, quirk)
        if quirk <= 2e-5:
            assert err <= TOL_REFERENCE, (name, err)
    print("worst rel-L2 vs reference goldens:", worst)


@pytest.fixture
def reference_growth(plx):
    """plx_tune("reference_growth", 1) for the test's builds: the reference CPU path's table-growth quirk replayed."""
    from simplex_gp_amd import _native as nv
    nv.check(nv.lib().plx_tune(b"reference_growth", 1), "plx_tune")
    yield
    nv.check(nv.lib().plx_tune(b"reference_growth", 0), "plx_tune")


def test_reference_growth_replay_vs_reference_goldens(plx, small, reference_growth):
    """With plx_tune("reference_growth", 1) the boundary call equals the REFERENCE's own output on every golden case at the
    north star's 1e-4 -- no quirk term -- and the replay reports the reference's vertex count (hashTable.size(), duplicates
    included).  cloud_grow_quirk_n2000_d8 is the case built to hit a doubling."""
    z, names = small
    worst = 0.0
    for name in names:
        ref, taps, src, gold, m_ref = (z[f"{name}/{k}"] for k in ("ref", "taps", "src", "out", "m"))
        lat = plx.Lattice().build(torch.from_numpy(ref).cuda(), taps)
        out = lat.apply(torch.from_numpy(src).cuda()).cpu().numpy()
        info = lat.reference_growth_info()
        assert info["replayed"] and not info["inexact"], (name, info)
        assert info["m_reference"] == int(m_ref), (name, info, int(m_ref))
        err = rel_l2(out, gold)
        worst = max(worst, err)
        assert err <= TOL_REFERENCE, (name, err, info)
        once = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), torch.from_numpy(taps)).cpu().numpy()
        assert rel_l2(once, gold) <= TOL_REFERENCE, name                 # the one-shot boundary call replays too
        lat.close()
    print("reference_growth: worst rel-L2 vs reference goldens:", worst)


@pytest.mark.parametrize("name", ["n1e5_d4_ell1.0", "n1e5_d4_ell0.25", "n1e5_d4_vd11_ell1.0",
                                  "n1e6_d8_ell1.0", "n1e6_d8_ell0.6931"])
def test_reference_growth_replay_large_probes(plx, golden_dir, name, reference_growth):
    """BASELINE.json config-2 / 3 shapes with the replay on: m == the reference's hashTable.size() and the output within 1e-4
    of the reference (its probes, and the reference-exact oracle over all rows) with NO quirk term -- n1e5_d4_ell0.25 and
    n1e6_d8_ell0.6931 are the two BASELINE inputs where the quirk alone moves the reference by more than 1e-4."""
    z = np.load(os.path.join(golden_dir, "filter_large.npz"))
    n, d, vd = (int(v) for v in z[f"{name}/shape"])
    g = torch.Generator().manual_seed(int(z[f"{name}/seed"]))
    x = torch.randn(n, d, generator=g)
    v = torch.randn(n, vd, generator=g)
    ref = (x / float(z[f"{name}/ell"])).contiguous()
    lat = plx.Lattice().build(ref.cuda(), z[f"{name}/taps"])
    out = lat.apply(v.cuda()).cpu().numpy()
    info = lat.reference_growth_info()
    print(name, info)
    assert info["replayed"] and not info["inexact"]
    assert info["m_reference"] == int(z[f"{name}/m"]), (info, int(z[f"{name}/m"]))
    stride = int(z[f"{name}/stride"])
    head, strided = z[f"{name}/out_head"], z[f"{name}/out_strided"]
    scale = float(z[f"{name}/out_l2"]) / np.sqrt(n * vd)
    assert np.linalg.norm(out[:512] - head) / (scale * np.sqrt(head.size)) <= TOL_REFERENCE
    assert np.linalg.norm(out[::stride] - strided) / (scale * np.sqrt(strided.size)) <= TOL_REFERENCE
    assert abs(np.linalg.norm(out.astype(np.float64)) / float(z[f"{name}/out_l2"]) - 1) <= TOL_REFERENCE
    exact = oracle.filter(v.numpy(), ref.numpy(), z[f"{name}/taps"])          # exact_mode: the reference, bit for bit
    err = rel_l2(out, exact)
    print(name, "rel-L2 vs the reference-exact oracle, replay on:", err)
    assert err <= TOL_REFERENCE
    lat.close()


QUIRK_SURVEY_SHAPES = [(16_599, 17, 0.6931), (45_730, 9, 0.6931), (48_827, 20, 0.6931), (100_000, 4, 0.25), (100_000, 4, 1.0),
                       (20_000, 8, 0.5), (200_000, 8, 1.0), (10_623, 18, 1.0), (60_000, 3, 0.1), (300_000, 8, 0.6931),
                       (2_000, 8, 0.5), (120_000, 6, 0.4)]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reference_growth_replay_quirk_survey(plx, reference_growth, seed):
    """The quirk survey's twelve shapes x three seeds (tests/checks/quirk_survey.py: the reference moves its own output by
    0 ... 2.9e-3 on them) with the replay on: rel-L2 against the reference-exact oracle <= 1e-4 on every input, no quirk
    term, and the replay's vertex count equal to the reference's."""
    import bench
    worst, hit = 0.0, 0
    for n, d, ell in QUIRK_SURVEY_SHAPES:
        g = torch.Generator().manual_seed(1000 * seed + n % 997 + d)
        x = torch.randn(n, d, generator=g)
        v = torch.randn(n, 1, generator=g)
        ref = (x / ell).contiguous()
        taps = bench.RBF1
        exact, m_exact = oracle.filter(v.numpy(), ref.numpy(), taps, return_m=True)
        oracle.set_exact_mode(False)
        clean = oracle.filter(v.numpy(), ref.numpy(), taps)
        oracle.set_exact_mode(True)
        lat = plx.Lattice().build(ref.cuda(), taps)
        out = lat.apply(v.cuda()).cpu().numpy()
        info = lat.reference_growth_info()
        quirk = rel_l2(clean, exact)
        err = rel_l2(out, exact)
        hit += quirk > TOL_REFERENCE
        worst = max(worst, err)
        assert info["replayed"] and not info["inexact"], (n, d, ell, info)
        assert info["m_reference"] == m_exact, (n, d, ell, info, m_exact)
        assert err <= TOL_REFERENCE, (n, d, ell, err, quirk, info)
        lat.close()
    print(f"seed {seed}: worst rel-L2 vs the reference with the replay on {worst:.2e}; inputs where the quirk alone exceeds 1e-4: {hit}")


def test_reference_growth_event_replay_equals_full_replay(plx):
    """plx_tune("reference_growth", 1) replays EVENTS (the m first-touch creations, the stale probe behind each doubling and
    every lookup of the keys such a probe has touched: O(m) host work); 2 runs all N (d+1) lookups against the table model
    (the round-5 form).  Same findings -- the reference's entry count, dropped lookups, invisible vertices, the blur-time
    miss -- and bit-identical outputs on the quirk survey's shapes and on the BASELINE lengthscale where the quirk exceeds
    1e-4 (N = 1e6, d = 8, l = 0.6931); the event form's wall time is printed beside the full form's."""
    import time
    import bench
    from simplex_gp_amd import _native as nv
    shapes = [(n, d, ell, 7) for n, d, ell in QUIRK_SURVEY_SHAPES] + [(1_000_000, 8, 0.6931, 1234), (1_000_000, 8, 1.0, 1234)]
    try:
        for n, d, ell, seed in shapes:
            g = torch.Generator().manual_seed(seed + n % 997 + d)
            x = torch.randn(n, d, generator=g)
            v = torch.randn(n, 1, generator=g).cuda()
            ref = (x / ell).contiguous().cuda()
            res = {}
            for mode in (2, 1):
                nv.check(nv.lib().plx_tune(b"reference_growth", mode), "plx_tune")
                lat = plx.Lattice().build(ref, bench.RBF1)                  # (sizes the buffers; the timed build follows)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lat.build(ref, bench.RBF1)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                res[mode] = (lat.reference_growth_info(), lat.apply(v).clone(), ms, lat.m)
                lat.close()
            nv.check(nv.lib().plx_tune(b"reference_growth", 0), "plx_tune")
            lat = plx.Lattice().build(ref, bench.RBF1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat.build(ref, bench.RBF1)
            torch.cuda.synchronize()
            plain_ms = (time.perf_counter() - t0) * 1e3
            lat.close()
            (full, out_full, ms_full, m), (ev, out_ev, ms_ev, _) = res[2], res[1]
            print(f"n={n} d={d} l={ell} m={m}: build {plain_ms:.2f} ms plain, +{ms_ev - plain_ms:.1f} ms event replay, "
                  f"+{ms_full - plain_ms:.1f} ms full replay; {ev}")
            assert ev == full, (n, d, ell, ev, full)
            assert torch.equal(out_ev, out_full), (n, d, ell)
            if n == 1_000_000 and ell < 1.0:
                assert ms_ev - plain_ms <= 100.0, (ms_ev, plain_ms)           # (measured 45-60 ms; the round-5 verdict asked for <= 60)
    finally:
        nv.check(nv.lib().plx_tune(b"reference_growth", 0), "plx_tune")


@pytest.mark.parametrize("name", ["n1e5_d4_ell1.0", "n1e5_d4_ell0.25", "n1e5_d4_vd11_ell1.0",
                                  "n1e6_d8_ell1.0", "n1e6_d8_ell0.6931"])
def test_large_vs_reference_probes(plx, golden_dir, name):
    """BASELINE.json config-2/3 shapes against probes of the reference output."""
    z = np.load(os.path.join(golden_dir, "filter_large.npz"))
    n, d, vd = (int(v) for v in z[f"{name}/shape"])
    g = torch.Generator().manual_seed(int(z[f"{name}/seed"]))
    x = torch.randn(n, d, generator=g)
    v = torch.randn(n, vd, generator=g)
    ref = (x / float(z[f"{name}/ell"])).contiguous()
    assert np.array_equal(ref[:8].numpy(), z[f"{name}/ref_head"])
    lat = plx.Lattice().build(ref.cuda(), z[f"{name}/taps"])
    out = lat.apply(v.cuda()).cpu().numpy()
    m_ref = int(z[f"{name}/m"])
    # the reference appends at most one duplicate vertex per table doubling (quirk)
    assert 0 <= m_ref - lat.m <= 12, (lat.m, m_ref)
    stride = int(z[f"{name}/stride"])
    head, strided = z[f"{name}/out_head"], z[f"{name}/out_strided"]
    scale = float(z[f"{name}/out_l2"]) / np.sqrt(n * vd)          # rms of the reference output
    # probes: error relative to the rms output level
    e_head = np.linalg.norm(out[:512] - head) / (scale * np.sqrt(head.size))
    e_str = np.linalg.norm(out[::stride] - strided) / (scale * np.sqrt(strided.size))
    l2 = np.linalg.norm(out.astype(np.float64))
    print(name, "m", lat.m, "m_ref", m_ref, "probe err", e_head, e_str, "l2 ratio", l2 / float(z[f"{name}/out_l2"]))
    # full output against the duplicate-free oracle (the lattice the HIP path builds)
    oracle.set_exact_mode(False)
    clean, m_clean = oracle.filter(v.numpy(), ref.numpy(), z[f"{name}/taps"], return_m=True)
    oracle.set_exact_mode(True)
    assert lat.m == m_clean
    err_clean = rel_l2(out, clean)
    print(name, "rel-L2 vs duplicate-free oracle", err_clean)
    assert err_clean <= TOL_ORACLE
    # against the reference itself: 1e-4, except where the reference's own hash-growth quirk (one orphaned vertex per
    # table doubling, h:105 vs h:61-63) moves ITS output by more than that.  The widening is measured here, with the
    # oracle in both modes on this very input -- never a literal: tol = max(1e-4, quirk + 1e-5).
    exact = oracle.filter(v.numpy(), ref.numpy(), z[f"{name}/taps"])          # exact_mode: the reference, bit for bit
    quirk = rel_l2(clean, exact)
    assert np.linalg.norm(exact[:512] - head) == 0.0                           # the oracle IS the reference here
    tol = max(TOL_REFERENCE, quirk + TOL_ORACLE)
    print(name, "reference quirk (oracle clean vs exact)", quirk, "-> tolerance", tol)
    assert rel_l2(out, exact) <= tol
    assert e_head <= tol and e_str <= tol
    assert abs(l2 / float(z[f"{name}/out_l2"]) - 1) <= tol
    lat.close()


def test_full_size_properties(plx):
    """N=1e6, d=8 (BASELINE.json metric shape): size-independent properties."""
    n, d = 1_000_000, 8
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(n, d, generator=g).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice().build(x, taps)
    a = torch.randn(n, 1, generator=g).cuda()
    b = torch.randn(n, 1, generator=g).cuda()
    Ka, Kb = lat.apply(a).clone(), lat.apply(b).clone()
    # determinism: same lattice, same input -> same bits
    assert torch.equal(lat.apply(a), Ka)
    # linearity
    Kab = lat.apply(2.0 * a - 3.0 * b)
    assert rel_l2(Kab.cpu().numpy(), (2.0 * Ka - 3.0 * Kb).cpu().numpy()) <= 1e-5
    # multi-column == column by column
    Kcat = lat.apply(torch.cat([a, b], 1))
    assert rel_l2(Kcat[:, :1].cpu().numpy(), Ka.cpu().numpy()) <= 1e-6
    assert rel_l2(Kcat[:, 1:].cpu().numpy(), Kb.cpu().numpy()) <= 1e-6
    # splat conserves mass: sum_v (S^T 1)_v = n  (barycentric weights sum to 1)
    ones = torch.ones(n, 1, device="cuda")
    assert abs(lat.splat(ones).double().sum().item() / n - 1) <= 1e-5
    # approximate symmetry of K: <a, K b> ~ <K a, b>   (viz_mvm.ipynb:150 reports 2e-2 asymmetry)
    lhs, rhs = (a * Kb).sum().item(), (Ka * b).sum().item()
    assert abs(lhs - rhs) <= 0.05 * max(abs(lhs), abs(rhs))
    # rebuilding on the same handle gives the same lattice and the same bits
    m0 = lat.m
    lat.build(x, taps)
    assert lat.m == m0 and torch.equal(lat.apply(a), Ka)
    lat.close()


def test_boundary_errors(plx):
    src = torch.randn(10, 2).cuda()
    ref = torch.randn(10, 3).cuda()
    taps = torch.tensor([0.5, 1.0, 0.5])
    with pytest.raises(ValueError):
        plx.filter(src[:5], ref, taps)                       # row mismatch (py:84-85)
    with pytest.raises(TypeError):
        plx.filter(src.double(), ref, taps)                  # CPU path is fp32 only (h:277-278)
    with pytest.raises(ValueError):
        plx.filter(src.cpu(), ref.cpu(), taps)               # no CPU fallback
    with pytest.raises(ValueError):
        plx.filter(src, ref, torch.tensor([0.5, 1.0]))       # even tap count
    from simplex_gp_amd._native import PlxError
    with pytest.raises(PlxError):
        plx.filter(src, ref * 1e6, taps)                     # int16 key overflow is an error, not UB
    with pytest.raises(PlxError):
        plx.filter(src, torch.full_like(ref, float("nan")), taps)
    # non-contiguous src is accepted (py:95 passes it as is)
    big = torch.randn(10, 4).cuda()
    out = plx.filter(big[:, ::2], ref, taps)
    assert rel_l2(out.cpu().numpy(), oracle.filter(big[:, ::2].cpu().numpy(), ref.cpu().numpy(), taps.numpy())) <= 1e-5
    # the handle survives an error
    out2 = plx.filter(src, ref, taps)
    assert torch.isfinite(out2).all()


def test_orders_and_dims(plx):
    """Every compiled dimension builds; taps of order 0..4."""
    rng = np.random.default_rng(0)
    for d in list(range(1, 33)):
        n = 300
        ref = rng.standard_normal((n, d)).astype(np.float32)
        src = rng.standard_normal((n, 2)).astype(np.float32)
        taps = np.array([0.3, 1.0, 0.3], np.float32)
        out = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), taps).cpu().numpy()
        assert rel_l2(out, oracle.filter(src, ref, taps)) <= TOL_ORACLE, d
    ref = rng.standard_normal((500, 3)).astype(np.float32) * 2
    src = rng.standard_normal((500, 5)).astype(np.float32)
    for taps in ([1.0], [0.5, 1, 0.5], [0.1, 0.5, 1, 0.5, 0.1], [0.02, 0.17, 0.64, 1, 0.64, 0.17, 0.02],
                 [0.01, 0.05, 0.2, 0.6, 1, 0.6, 0.2, 0.05, 0.01]):
        taps = np.array(taps, np.float32)
        out = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), taps).cpu().numpy()
        assert rel_l2(out, oracle.filter(src, ref, taps)) <= TOL_ORACLE, len(taps)


def test_one_shot_plx_filter_through_raw_abi(plx):
    """plx_filter(scratch=NULL, ...) = the reference's filter(): builds, applies, leaves nothing behind."""
    import ctypes
    from simplex_gp_amd import _native as nv
    rng = np.random.default_rng(3)
    n, d, vd = 4000, 3, 2
    ref = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    src = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).cuda()
    out = torch.empty_like(src)
    taps = np.array([0.5, 1.0, 0.5], np.float32)
    rc = nv.lib().plx_filter(None, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(ref.data_ptr()), n, d, vd,
                             taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 3, ctypes.c_void_p(out.data_ptr()),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, nv.lib().plx_last_error()
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().numpy(), oracle.filter(src.cpu().numpy(), ref.cpu().numpy(), taps)) <= TOL_ORACLE
    # bad arguments come back as codes with a message, never as a crash
    assert nv.lib().plx_filter(None, None, ctypes.c_void_p(ref.data_ptr()), n, d, vd,
                               taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 3, ctypes.c_void_p(out.data_ptr()), None) != 0
    assert nv.lib().plx_filter(None, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(ref.data_ptr()), n, 99, vd,
                               taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 3, ctypes.c_void_p(out.data_ptr()), None) == 4


def test_wide_right_hand_sides(plx):
    """vd = 2L(1+d) columns as in the backward pass (py:113-119): column tiles and padding."""
    rng = np.random.default_rng(5)
    n, d = 3000, 4
    ref = rng.standard_normal((n, d)).astype(np.float32)
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    lat = plx.Lattice().build(torch.from_numpy(ref).cuda(), taps)
    for vd in (2, 5, 12, 13, 30, 64, 65, 68, 101, 110, 124, 128, 198, 300):      # >= 65 columns (17 chunks) take the row-parallel splat
        src = rng.standard_normal((n, vd)).astype(np.float32)
        out = lat.apply(torch.from_numpy(src).cuda()).cpu().numpy()
        assert out.shape == (n, vd)
        oracle.set_exact_mode(False)
        want = oracle.filter(src, ref, taps)
        oracle.set_exact_mode(True)
        assert rel_l2(out, want) <= TOL_ORACLE, vd


def test_random_small_shapes(plx):
    """Randomised shapes, scales, orders and degenerate clouds against the oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(n=st.integers(1, 400), d=st.integers(1, 7), vd=st.integers(1, 6), order=st.integers(0, 3),
           scale=st.sampled_from([0.05, 0.5, 1.0, 3.0, 20.0]), kind=st.sampled_from(["normal", "grid", "dup", "line"]),
           seed=st.integers(0, 10**6))
    def run(n, d, vd, order, scale, kind, seed):
        rng = np.random.default_rng(seed)
        if kind == "normal":
            ref = rng.standard_normal((n, d))
        elif kind == "grid":                       # many exact ties in the rounding / ranking
            ref = rng.integers(-3, 4, (n, d)).astype(np.float64) * 0.5
        elif kind == "dup":                        # few distinct points, many duplicates
            ref = rng.standard_normal((max(1, n // 20), d))[rng.integers(0, max(1, n // 20), n)]
        else:                                      # points on a line
            ref = np.outer(rng.standard_normal(n), rng.standard_normal(d))
        ref = (ref * scale).astype(np.float32)
        src = rng.standard_normal((n, vd)).astype(np.float32)
        taps = np.array([0.1, 0.3, 0.6, 1.0, 0.6, 0.3, 0.1][3 - order: 4 + order], np.float32)
        out = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), taps).cpu().numpy()
        oracle.set_exact_mode(False)
        want, m = oracle.filter(src, ref, taps, return_m=True)
        oracle.set_exact_mode(True)
        err = np.linalg.norm(out - want) / max(np.linalg.norm(want), 1e-20)
        # hundreds of points collapsing into one or two vertices with random signs: the oracle adds them one
        # by one in fp32, the scan adds them as a tree, and cancellation amplifies the difference to ~1e-5;
        # 5e-5 keeps a 2x margin to the 1e-4 north-star bound
        assert err <= 5e-5, (n, d, vd, order, scale, kind, seed, err)

    run()


@pytest.mark.parametrize("order", [1, 2, 3])
def test_compacted_neighbour_table_equals_dense(plx, order):
    """The wave-prefix compacted neighbour table (used on sparse lattices at vd = 1) gives bit-identical blur
    results to the dense [d+1][2r][m] table, including vertex counts that are not multiples of 4 / 256."""
    from simplex_gp_amd import _native as nv
    rng = np.random.default_rng(50 + order)
    half = {1: [0.34608543], 2: [0.08263808, 0.53616077], 3: [0.01831428, 0.16900772, 0.64117509]}[order]
    taps = np.asarray(half + [1.0] + half[::-1], np.float32)
    lib = nv.lib()
    try:
        for n, d, scale in [(3001, 3, 4.0), (20000, 5, 3.0), (777, 8, 2.0), (50000, 2, 30.0)]:
            ref = torch.from_numpy((rng.standard_normal((n, d)) * scale).astype(np.float32)).cuda()
            src = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).cuda()
            outs, ms = [], []
            for mode in (0, 2):
                nv.check(lib.plx_tune(b"compact_nbr", mode), "plx_tune")
                lat = plx.Lattice().build(ref, taps)
                vals = lat.splat(src)
                res = lat.blur(vals.clone(), vd=1)
                outs.append(res.cpu().numpy().copy())
                ms.append(lat.m)
                lat.close()
            assert ms[0] == ms[1]
            assert np.array_equal(outs[0], outs[1]), (n, d, order)
    finally:
        nv.check(lib.plx_tune(b"compact_nbr", 1), "plx_tune")


@pytest.mark.parametrize("vd", [1, 3, 7, 12, 40, 70, 101, 130])
def test_long_vertex_rows_every_splat_kernel(plx, vd):
    """Clouds whose points share a handful of simplices: every vertex row is hundreds to thousands of corners long,
    so rows span many lane-group runs, waves and workgroups (in-wave segmented scan + head / tail partials + fix-up)
    in each of the splat kernels (scan: vd 1-4, lane groups: 5-64, wide: >= 65 -- with idle lanes up to 124)."""
    rng = np.random.default_rng(200 + vd)
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    cases = []
    n, d = 6001, 5
    same = np.tile(rng.standard_normal((1, d)).astype(np.float32), (n, 1))              # one simplex: rows of n corners
    cases.append(same)
    few = rng.standard_normal((7, d)).astype(np.float32)[rng.integers(0, 7, n)]          # 7 distinct positions
    cases.append(few + 1e-4 * rng.standard_normal((n, d)).astype(np.float32))
    cases.append((rng.standard_normal((3000, 2)) * 0.05).astype(np.float32))            # a tight 2-D blob
    oracle.set_exact_mode(False)
    try:
        for ref in cases:
            src = rng.standard_normal((ref.shape[0], vd)).astype(np.float32)
            want = oracle.filter(src, ref, taps)
            got = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), torch.from_numpy(taps)).cpu().numpy()
            assert rel_l2(got, want) <= TOL_ORACLE, (vd, ref.shape)
    finally:
        oracle.set_exact_mode(True)


@pytest.mark.parametrize("ntaps", [1, 3, 5, 7])
def test_every_tap_order_at_every_row_width(plx, ntaps):
    """Order 0 (a single tap: the blur is the identity times c) up to order 3, through the single-column, narrow,
    general and wide blur kernels (a fuzz run caught order 0 being sent to an order-3 instantiation at >= 125 columns)."""
    rng = np.random.default_rng(300 + ntaps)
    taps = np.array([0.1, 0.3, 0.6, 1.0, 0.6, 0.3, 0.1][3 - ntaps // 2: 4 + ntaps // 2], np.float32)
    n, d = 1500, 3
    ref = (rng.standard_normal((n, d)) * 2.0).astype(np.float32)
    oracle.set_exact_mode(False)
    try:
        for vd in (1, 3, 7, 17, 40, 66, 110, 130):
            src = rng.standard_normal((n, vd)).astype(np.float32)
            want = oracle.filter(src, ref, taps)
            got = plx.filter(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), torch.from_numpy(taps)).cpu().numpy()
            assert rel_l2(got, want) <= TOL_ORACLE, (ntaps, vd)
    finally:
        oracle.set_exact_mode(True)


@pytest.mark.parametrize("block_e", [16, 24])
def test_block_tables_equal_csr_path(plx, block_e):
    """vd = 1 through the block tables (plx_block.hip: block-local splat + per-vertex combine, LDS-staged slice; blocks of
    256 * block_e corners) against the vertex-sorted CSR kernels and the oracle: caller row order, lattice row order, the
    affine epilogue, an owned row range (sharded structure) and shapes where blocks are ragged (n not a multiple of the
    block, d + 1 not a divisor of the block's corner count)."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    rng = np.random.default_rng(77)
    shapes = [(50, 1, 1.0), (4099, 2, 1.0), (30011, 4, 1.0), (20000, 8, 2.0), (9000, 18, 4.0), (6001, 5, 1e3), (100003, 3, 0.5)]
    try:
        for n, d, scale in shapes:
            ref = (rng.standard_normal((n, d)) / scale).astype(np.float32)
            src = rng.standard_normal((n, 1)).astype(np.float32)
            oracle.set_exact_mode(False)
            want = oracle.filter(src, ref, taps)
            oracle.set_exact_mode(True)
            x, s = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
            nv.check(lib.plx_tune(b"block_path", 0), "plx_tune")
            a = plx.Lattice().build(x, taps)
            out_a = a.apply(s).clone()
            assert a.block_rows == 0
            va = a.splat(s).clone()
            nv.check(lib.plx_tune(b"block_path", 2), "plx_tune")
            nv.check(lib.plx_tune(b"block_e", block_e), "plx_tune")
            b = plx.Lattice().build(x, taps)
            assert b.block_rows == 0                     # the tables are built by their first user ...
            assert b.prepare(1).block_rows > 0 and b.m == a.m      # ... or by plx_prepare
            out_b = b.apply(s).clone()
            assert "splat_block_kernel" in b.stage_kernels()["splat"] and b.stage_kernels()["slice"][0] == "slice_block_kernel"
            assert rel_l2(out_b.cpu().numpy(), want) <= TOL_ORACLE, (n, d)
            assert rel_l2(out_b.cpu().numpy(), out_a.cpu().numpy()) <= 5e-6, (n, d)          # two fixed summation trees
            # stage by stage: same vertex numbering in both builds
            vb = b.splat(s)
            assert rel_l2(vb.cpu().numpy(), va.cpu().numpy()) <= 5e-6
            blurred = a.blur(va.clone(), vd=1)
            assert torch.equal(b.slice(blurred, vd=1), a.slice(blurred, vd=1))        # same arithmetic, same order
            assert torch.equal(b.apply(s), out_b)                                       # reproducible bits
            # the combine kernel that numbers vertices by counting row ends against the one that reads their ids
            b.tune("block_dense_combine", 0)
            assert torch.equal(b.splat(s), vb)
            b.tune("block_dense_combine", 1)
            # the switches are per lattice: `a` was built under block_path = 0 and stays on the CSR path although the
            # process default has been 2 since
            assert torch.equal(a.apply(s), out_a) and a.block_rows == 0 and "block" not in a.stage_kernels()["splat"][0]
            # affine epilogue and lattice row order
            ss = torch.tensor([0.7, 0.3], device="cuda")
            assert rel_l2(b.apply_affine(s, ss).cpu().numpy(), (0.7 * out_b + 0.3 * s).cpu().numpy()) <= 1e-6
            b.set_lattice_row_order(True)
            out_l = b.from_lattice_order(b.apply(b.to_lattice_order(s)))
            b.set_lattice_row_order(False)
            assert torch.equal(out_l, out_b)
            # multi-column right-hand sides on a lattice with block tables use the CSR built on demand, next to them
            s3 = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda()
            assert rel_l2(b.apply(s3).cpu().numpy(), a.apply(s3).cpu().numpy()) <= 1e-6
            assert torch.equal(b.apply(s), out_b)                                       # and the single-column tables still stand
            # an owned row range: per-shard splats add up, per-shard slices tile the output
            if n >= 4099:
                from simplex_gp_amd.distributed import shard_bounds
                total, parts = None, []
                for r in range(3):
                    lo, hi = shard_bounds(n, 3, r)
                    lat = plx.Lattice().build(x, taps, shard=(r, 3))
                    part = lat.splat(s[lo:hi])
                    assert lat.block_rows > 0 and "block" in lat.stage_kernels()["splat"][0]
                    total = part.clone() if total is None else total + part
                    parts.append((lat, lo, hi))
                got = torch.empty_like(out_b)
                for lat, lo, hi in parts:
                    got[lo:hi] = lat.slice(lat.blur(total.clone(), vd=1), vd=1)
                    lat.close()
                assert rel_l2(got.cpu().numpy(), want) <= TOL_ORACLE, (n, d)
            a.close()
            b.close()
    finally:
        nv.check(lib.plx_tune(b"block_path", 1), "plx_tune")
        nv.check(lib.plx_tune(b"block_e", 0), "plx_tune")


def test_blur_axis_pairs_equal_single_axis_passes(plx):
    """Order-1, vd = 1 blur with two axes per launch (composite neighbour table) against one axis per launch: the
    same fp32 operations in the same order, so the same bits; d + 1 even and odd, dense and sparse neighbourhoods."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    rng = np.random.default_rng(91)
    try:
        for n, d, scale in [(3000, 1, 1.0), (20000, 2, 1.0), (30000, 3, 0.5), (50000, 8, 1.0), (20000, 5, 0.2), (20011, 4, 1.0)]:
            ref = (rng.standard_normal((n, d)) / scale).astype(np.float32)
            src = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).cuda()
            x = torch.from_numpy(ref).cuda()
            outs, blurs = [], []
            for mode in (0, 2):
                nv.check(lib.plx_tune(b"blur_fuse", mode), "plx_tune")
                nv.check(lib.plx_tune(b"blur_small", 0), "plx_tune")      # small lattices: per-axis kernels, not the LDS one
                lat = plx.Lattice().build(x, taps)
                vals = lat.splat(src)
                blurs.append(lat.blur(vals.clone(), vd=1).clone())
                outs.append(lat.apply(src).clone())
                names = lat.stage_kernels()["blur_axis"]
                assert ("blur_pair_v1_kernel" in names) == (mode == 2 and "blur_axis_compact_kernel" not in names), (mode, names)
                lat.close()
            assert torch.equal(blurs[0], blurs[1]), (n, d)
            assert torch.equal(outs[0], outs[1]), (n, d)
            oracle.set_exact_mode(False)
            want = oracle.filter(src.cpu().numpy(), ref, taps)
            oracle.set_exact_mode(True)
            assert rel_l2(outs[1].cpu().numpy(), want) <= TOL_ORACLE
    finally:
        nv.check(lib.plx_tune(b"blur_fuse", 1), "plx_tune")
        nv.check(lib.plx_tune(b"blur_small", 1), "plx_tune")


def test_vertex_order_morton_is_a_relabelling(plx):
    """vertex_order = 2 numbers the vertices along the Morton curve of the blur-axis coordinates instead of by first
    touch: the same vertex set under other ids, so the filter output agrees to fp32 rounding (the per-vertex sums see
    their block rows in another order), the keys agree as sets, and the oracle still bounds the result."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    rng = np.random.default_rng(97)
    try:
        for n, d, scale, vd in [(3000, 1, 1.0, 1), (20000, 2, 1.0, 3), (50000, 8, 1.0, 1), (20000, 5, 0.2, 1),
                                (20011, 4, 1.0, 12), (30000, 12, 1.0, 1), (10000, 20, 1.0, 2)]:
            ref = (rng.standard_normal((n, d)) / scale).astype(np.float32)
            src = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).cuda()
            x = torch.from_numpy(ref).cuda()
            outs, keys = [], []
            for mode in (0, 2):
                nv.check(lib.plx_tune(b"vertex_order", mode), "plx_tune")
                lat = plx.Lattice().build(x, taps)
                outs.append(lat.apply(src).clone())
                assert lat.stage_kernels()["vertex_order"] == [["first_touch", None, "morton"][mode]]
                k = lat.export(nv.ARRAY_KEYS)
                keys.append(k[np.lexsort(k.T[::-1])])
                lat.close()
            assert np.array_equal(keys[0], keys[1]), (n, d)
            assert rel_l2(outs[1].cpu().numpy(), outs[0].cpu().numpy()) <= 1e-6, (n, d)
            oracle.set_exact_mode(False)
            want = oracle.filter(src.cpu().numpy(), ref, taps)
            oracle.set_exact_mode(True)
            assert rel_l2(outs[1].cpu().numpy(), want) <= TOL_ORACLE
    finally:
        nv.check(lib.plx_tune(b"vertex_order", 1), "plx_tune")


def test_blur_axis_pairs_for_rows_equal_single_axis_passes(plx):
    """Order-1 blur of 2..16 columns with two axes per launch (blur_pair_narrow_kernel over the composite neighbour table)
    against one axis per launch: the same fp32 operations in the same order, so the same bits; rows of 2, 3 and 4
    chunks, d + 1 even and odd, dense and sparse neighbourhoods, both vertex numberings."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    rng = np.random.default_rng(93)
    try:
        for n, d, scale, vd, vo in [(3000, 1, 1.0, 5, 1), (20000, 2, 1.0, 8, 1), (30000, 3, 0.5, 12, 2), (50000, 8, 1.0, 11, 2),
                                    (20000, 5, 0.2, 16, 1), (20011, 4, 1.0, 7, 2)]:
            nv.check(lib.plx_tune(b"vertex_order", vo), "plx_tune")
            ref = (rng.standard_normal((n, d)) / scale).astype(np.float32)
            src = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).cuda()
            x = torch.from_numpy(ref).cuda()
            outs, blurs = [], []
            for mode in (0, 1):
                nv.check(lib.plx_tune(b"blur_fuse_vec", mode), "plx_tune")
                lat = plx.Lattice().build(x, taps)
                vals = lat.splat(src)
                blurs.append(lat.blur(vals.clone(), vd=vd).clone())
                outs.append(lat.apply(src).clone())
                names = lat.stage_kernels()["blur_axis"]
                assert ("blur_pair_narrow_kernel" in names) == (mode == 1), (mode, names)
                lat.close()
            assert torch.equal(blurs[0], blurs[1]), (n, d, vd)
            assert torch.equal(outs[0], outs[1]), (n, d, vd)
            oracle.set_exact_mode(False)
            want = oracle.filter(src.cpu().numpy(), ref, taps)
            oracle.set_exact_mode(True)
            assert rel_l2(outs[1].cpu().numpy(), want) <= TOL_ORACLE
    finally:
        nv.check(lib.plx_tune(b"blur_fuse_vec", 1), "plx_tune")
        nv.check(lib.plx_tune(b"vertex_order", 1), "plx_tune")


def test_one_shot_filter_skips_what_only_pays_over_many_mvms(plx):
    """plx_filter (the reference's build-per-call contract) builds for a single MVM: no vertex renumbering, no axis-pair
    tables; same output as build() + apply() to fp32 rounding, and the lattice it leaves behind is a normal one."""
    rng = np.random.default_rng(5)
    n, d = 300000, 8
    ref = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    for vd in (1, 6):
        src = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).cuda()
        many = plx.Lattice().build(ref, taps)
        want = many.apply(src)
        assert many.m >= 65536 and many.stage_kernels()["vertex_order"] == ["morton"]
        once = plx.Lattice()
        got = once.filter_once(src, ref, taps)
        names = once.stage_kernels()
        assert names["vertex_order"] == ["first_touch"]
        assert not any("pair" in k for k in names["blur_axis"]), names
        assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) <= 1e-6
        assert rel_l2(once.apply(src).cpu().numpy(), want.cpu().numpy()) <= 1e-6     # still a usable lattice
        assert rel_l2(plx.filter(src, ref, torch.from_numpy(taps)).cpu().numpy(), want.cpu().numpy()) <= 1e-6
        once.build(ref, taps)                                                          # and build() resets the mode
        assert once.stage_kernels()["vertex_order"] == ["morton"]
        many.close(); once.close()


def test_tables_are_built_by_their_first_user(plx):
    """plx_build builds the lattice structure only; the splat / slice tables (block tables, their vertex-sorted half, the
    vertex-sorted CSR) are built by the first MVM that reads them, or by plx_prepare.  Whatever the order of first use --
    including multi-column, single-column, multi-column on one lattice, which once shared a sort buffer between the
    block build and the CSR -- the results equal those of a lattice prepared up front."""
    rng = np.random.default_rng(23)
    n, d = 200000, 8
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    x2 = torch.from_numpy((rng.standard_normal((n, d)) / 0.9).astype(np.float32)).cuda()
    v1 = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).cuda()
    v11 = torch.from_numpy(rng.standard_normal((n, 11)).astype(np.float32)).cuda()
    v20 = torch.from_numpy(rng.standard_normal((n, 20)).astype(np.float32)).cuda()
    fresh = plx.Lattice().build(x2, taps)
    assert fresh.block_rows == 0
    fresh.prepare(1).prepare(11).prepare(20)
    assert fresh.block_rows > 0
    want1, want11, want20 = fresh.apply(v1).clone(), fresh.apply(v11).clone(), fresh.apply(v20).clone()
    for order in ((v20, v1, v11, v20), (v11, v1, v20, v11), (v1, v20, v11, v1)):
        lat = plx.Lattice().build(x2, taps)
        for v in order:
            want = {1: want1, 11: want11, 20: want20}[v.shape[1]]
            assert torch.equal(lat.apply(v), want), [t.shape[1] for t in order]
        assert lat.block_rows == fresh.block_rows
        lat.build(x2, taps)                            # a rebuild starts over: nothing of the old tables is trusted
        assert lat.block_rows == 0 and torch.equal(lat.apply(v11), want11) and torch.equal(lat.apply(v1), want1)
        lat.close()
    # timing: a build no longer contains the tables; prepare() does
    lat = plx.Lattice()
    lat.set_timing(True)
    lat.build(x2, taps); lat.build(x2, taps)
    assert lat.build_times_ms()["csr"] < 0.05
    lat.set_timing(False)
    # one-shot filters: results unchanged (a multi-column one-shot call keeps to the CSR kernels)
    once = plx.Lattice().filter_once(v11, x2, taps)
    assert rel_l2(once.cpu().numpy(), want11.cpu().numpy()) <= 1e-6
    assert rel_l2(plx.Lattice().filter_once(v1, x2, taps).cpu().numpy(), want1.cpu().numpy()) <= 1e-6
    lat.close(); fresh.close()


def test_first_touch_splat_equals_sorted_corner_path(plx):
    """vd = 1 on lattices where (almost) every corner owns its vertex -- plx_first.hip: the first-touch corner of every
    vertex stores its product (as one contiguous run per workgroup under first-touch numbering, scattered otherwise), the
    few remaining corners are added from a short list sorted by vertex -- against the vertex-sorted segmented scan and the
    oracle: fine clouds, clouds with exact duplicates (extras), both row orders, and a heavy tail (hundreds of points on one
    spot) that must send the lattice back to the sorted-corner path."""
    from simplex_gp_amd import _native as nv
    lib = nv.lib()
    taps = np.array([0.34608543, 1.0, 0.34608543], np.float32)
    rng = np.random.default_rng(123)
    cases = []
    for n, d, scale in [(30011, 8, 0.25), (5000, 3, 0.02), (20000, 18, 0.5), (257, 2, 0.001), (100003, 5, 0.1)]:
        ref = (rng.standard_normal((n, d)) / scale).astype(np.float32)
        cases.append(("fine", ref, True))
        dup = ref.copy()
        dup[n // 2: n // 2 + n // 50] = dup[: n // 50]                       # 2 % exact duplicates: every corner of them is an extra
        cases.append(("duplicates", dup, True))
    heavy = (rng.standard_normal((20000, 4)) / 0.01).astype(np.float32)
    heavy[:300] = heavy[0]                                                    # 300 points on one spot: runs of 299 extras
    cases.append(("heavy tail", heavy, False))
    try:
        for name, ref, expect_first in cases:
            n, d = ref.shape
            src = rng.standard_normal((n, 1)).astype(np.float32)
            oracle.set_exact_mode(False)
            want = oracle.filter(src, ref, taps)
            oracle.set_exact_mode(True)
            x, s = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
            outs = {}
            for mode in (0, 1, 3):
                nv.check(lib.plx_tune(b"splat_first", mode), "plx_tune")
                lat = plx.Lattice().build(x, taps)
                out = lat.apply(s).clone()
                kern = lat.stage_kernels()["splat"]
                used = any("splat_first" in k for k in kern)
                assert used == (mode != 0 and expect_first), (name, n, d, mode, kern)
                if mode == 1 and expect_first:
                    assert "splat_first_seq_kernel" in kern          # first-touch numbering on these lattices: the contiguous store
                assert rel_l2(out.cpu().numpy(), want) <= TOL_ORACLE, (name, n, d, mode)
                assert torch.equal(lat.apply(s), out)                # reproducible bits
                vals = lat.splat(s).clone()
                lat.set_lattice_row_order(True)
                assert torch.equal(lat.from_lattice_order(lat.apply(lat.to_lattice_order(s))), out)
                assert torch.equal(lat.splat(lat.to_lattice_order(s)), vals)
                lat.set_lattice_row_order(False)
                # a multi-column MVM on the same lattice builds and uses the sorted corners next to it
                s3 = torch.cat([s, 2 * s, -s], 1).contiguous()
                o3 = lat.apply(s3)
                assert rel_l2(o3[:, :1].cpu().numpy(), want) <= TOL_ORACLE
                assert torch.equal(lat.apply(s), out)
                outs[mode] = (out, vals, lat.m)
                lat.close()
            assert outs[0][2] == outs[1][2] == outs[3][2]
            assert rel_l2(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy()) <= 2e-6      # two fixed summation orders
            assert torch.equal(outs[1][1], outs[3][1])                                    # contiguous and scattered stores: same sums
    finally:
        nv.check(lib.plx_tune(b"splat_first", 1), "plx_tune")
