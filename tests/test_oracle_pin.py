"""Pins the CPU oracle (oracle/loops_oracle.c): against the committed golden fixtures produced
by the reference's own host code (tests/golden/make_golden.py), against the known answers the
reference's unit tests hold, and -- when oracle/_ref is present -- live against the reference."""
import ctypes as C

import numpy as np
import pytest

from conftest import battery, load_golden
from oracle import oracle as O


def test_chesapeake_known_answers():
    g = load_golden("chesapeake.npz")
    # SURVEY App. D.1 / BASELINE.md C1: sum(y) = 1794, y[0..7] = 50,52,53,26,18,21,51,66
    assert int(g["rows"]) == 39 and g["indices"].size == 340
    assert g["y"].sum() == 1794 and list(g["y"][:8]) == [50, 52, 53, 26, 18, 21, 51, 66]
    assert list(g["x"][:8]) == [1, 10, 6, 2, 10, 6, 5, 5]
    y = O.spmv_f32(g["offsets"], g["indices"], g["values"], g["x"])
    assert np.array_equal(y, g["y"])
    assert np.array_equal(O.spmv_f64acc_f32(g["offsets"], g["indices"], g["values"], g["x"]), g["y_f64acc"])
    assert np.array_equal(O.xgen_int(39, 1, 10, 42), g["x"])


def test_xgen_golden():
    g = load_golden("xgen.npz")
    assert np.array_equal(O.xgen_int(4096, 1, 10, 42), g["x_1_10_42"])
    assert np.array_equal(O.xgen_int(4096, 0, 1, 7), g["x_0_1_7"])
    assert np.array_equal(O.xgen_int(4096, -5, 5, 12345), g["x_m5_5_12345"])
    assert [O.lib().oracle_hash(a) for a in (0, 1, 2, 41, 12345, 2**32 - 1)] == list(g["hash"])


def test_battery_golden():
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        assert np.array_equal(off, g[name + ".offsets"]) and np.array_equal(val, g[name + ".values"]), name
        for tag in ("int", "real"):
            x = g[f"{name}.x_{tag}"]
            assert np.array_equal(O.spmv_f32(off, idx, val, x), g[f"{name}.y_{tag}"]), (name, tag)
            assert np.array_equal(O.spmv_f64acc_f32(off, idx, val, x), g[f"{name}.y64_{tag}"]), (name, tag)
            assert np.array_equal(O.row_l1_f32(off, idx, val, x), g[f"{name}.l1_{tag}"]), (name, tag)
            assert np.array_equal(O.spmv_f32(off, idx, val, x, omp=True), g[f"{name}.y_{tag}"]), (name, tag)
        for R in (2, 3, 4):
            if f"{name}.bcsr{R}.offsets" in g:
                bo, bc, bv = O.csr_to_bcsr_f32(R, R, r, c, off, idx, val)
                assert np.array_equal(bo, g[f"{name}.bcsr{R}.offsets"]), (name, R)
                assert np.array_equal(bc, g[f"{name}.bcsr{R}.cols"]) and np.array_equal(bv, g[f"{name}.bcsr{R}.values"])
                # BCSR SpMV == CSR SpMV on exactly-summable input
                xi = np.zeros(((c + R - 1) // R) * R, np.float32)
                xi[:c] = g[name + ".x_int"]
                yb = O.bcsr_spmv_f32(R, R, r, bo, bc, bv, xi)
                assert np.allclose(yb, g[name + ".y_int"], rtol=1e-6, atol=1e-6), (name, R)


def test_layout_golden_and_reference_unit_test_answers():
    g = load_golden("layouts.npz")
    for name in ("csr4", "bcsr5", "one_row", "all_empty"):
        o = g[name + ".offsets"]
        nt, na = o.size - 1, int(o[-1])
        p = o.ctypes.data_as(C.c_void_p)
        for w in range(4):
            assert [O.lib().oracle_layout_csr(p, nt, na, w, t) for t in range(nt)] == list(g[name + ".csr"][w])
        assert [O.lib().oracle_layout_csr(p, nt, na, 4, a) for a in range(na)] == list(g[name + ".tile_of"])
        for K in (2, 4, 8, 16):
            T = O.lib().oracle_layout_flat(K, p, nt, na, 5, 0)
            assert T == g[f"{name}.flat{K}"].shape[1]
            for w in range(4):
                assert [O.lib().oracle_layout_flat(K, p, nt, na, w, t) for t in range(T)] == list(g[f"{name}.flat{K}"][w])
            assert [O.lib().oracle_layout_flat(K, p, nt, na, 4, a) for a in range(na)] == list(g[f"{name}.flat{K}.tile_of"])
            assert [O.lib().oracle_layout_flat(K, p, nt, na, 7, a) for a in range(na)] == list(g[f"{name}.flat{K}.base_tile_of"])
    # unittests/test_layout_csr.cu:23-52: offsets {0,2,2,5,7}
    o = np.array([0, 2, 2, 5, 7], np.int32)
    p = o.ctypes.data_as(C.c_void_p)
    assert [O.lib().oracle_layout_csr(p, 4, 7, 4, a) for a in (0, 1, 2, 4, 5, 6)] == [0, 0, 2, 2, 3, 3]
    assert [O.lib().oracle_layout_csr(p, 4, 7, 2, t) for t in range(4)] == [2, 0, 3, 2]
    # unittests/test_layout_flat_partitioner.cu:24-112: K=2 over 7 atoms -> 4 tiles (2,2,2,1)
    assert O.lib().oracle_layout_flat(2, p, 4, 7, 5, 0) == 4
    assert [O.lib().oracle_layout_flat(2, p, 4, 7, 2, t) for t in range(4)] == [2, 2, 2, 1]
    assert [O.lib().oracle_layout_flat(2, p, 4, 7, 4, a) for a in range(7)] == [a // 2 for a in range(7)]
    assert O.lib().oracle_layout_flat(2, p, 4, 7, 7, 2) == 2
    for nt, pitch in ((5, 3), (1, 7), (4, 1)):
        for w in range(4):
            assert [O.lib().oracle_layout_ell(nt, pitch, w, t) for t in range(nt)] == list(g[f"ell{nt}x{pitch}"][w])
        assert [O.lib().oracle_layout_ell(nt, pitch, 4, a) for a in range(nt * pitch)] == list(g[f"ell{nt}x{pitch}.tile_of"])
    for w in range(5):
        assert [O.lib().oracle_layout_coo(9, w, t) for t in range(9)] == list(g["coo9"][w])


def test_ceil_div_reference_unit_test_answers():
    # unittests/test_util_math.cu:22-67
    cd = O.lib().oracle_ceil_div
    assert cd(10, 5) == 2 and cd(11, 5) == 3 and cd(0, 7) == 0 and cd(1, 7) == 1 and cd(7, 7) == 1
    assert cd(2**63 - 1, 2) == 2**62 and cd(2**63 - 1, 2**63 - 1) == 1 and cd(2**63 - 1, 1) == 2**63 - 1


def test_diagonal_search_known_answers():
    # SURVEY App. A.1 table for offsets {0,2,2,5,7} (fixture of unittests/test_layout_csr.cu:26-31)
    te = np.array([2, 2, 5, 7], np.int32)
    got = [O.diag_search(d, te, 0, 4, 7) for d in range(13)]
    assert got == [(0, 0), (0, 1), (0, 2), (1, 2), (2, 2), (2, 3), (2, 4), (2, 5), (3, 5), (3, 6), (3, 7), (4, 7), (4, 7)]


@pytest.mark.parametrize("tile", [(256, 8), (128, 7), (4, 2), (256, 7)])
def test_schedule_assignments_are_exact_covers(tile):
    """Every schedule visits every atom exactly once and attributes it to its true row
    (what unittests/test_schedule_coverage.cu:54-112 checks for thread_mapped)."""
    tpb, ipt = tile
    for name, (r, c, off, idx, val) in battery().items():
        nnz = idx.size
        true_row = np.searchsorted(off[1:], np.arange(nnz), side="right")
        ts, owner, row, vis = O.merge_path_assign(off, tpb, ipt)
        assert (vis == 1).all() and np.array_equal(row, true_row), name
        coords = O.merge_path_coords(off, tpb, ipt)
        M = O.merge_path_num_tiles(r, nnz, tpb, ipt)
        assert coords.shape[0] == M + 1 and tuple(coords[0]) == (0, 0) and tuple(coords[-1]) == (r, nnz)
        assert (np.diff(coords[:, 0].astype(np.int64)) >= 0).all() and (np.diff(coords[:, 1].astype(np.int64)) >= 0).all()
        d = coords.astype(np.int64).sum(1)
        assert np.array_equal(d[:-1], np.arange(M) * tpb * ipt)
        # owner is monotone in atom order (contiguous even shares)
        assert (np.diff(owner) >= 0).all()
        for nthreads in (256, 1024, 7 * 256):
            tm, owner, row, vis = O.work_oriented_assign(off, nthreads)
            assert (vis == 1).all() and np.array_equal(row, true_row), name
        for G in (16, 64, 256):
            owner, row, vis = O.group_mapped_assign(off, G)
            assert (vis == 1).all() and np.array_equal(row, true_row), name
        y = O.merge_path_spmv_f32(off, idx, val, load_golden("battery.npz")[name + ".x_int"], tpb, ipt)
        assert np.allclose(y, load_golden("battery.npz")[name + ".y_int"], rtol=1e-6, atol=1e-5)


def test_rigorous_validator_semantics():
    # unittests/test_rigorous_validator.cu:85-144: identity -> 0 overruns, max_abs == 0; a corrupted
    # entry is flagged; cancellation rows stay within the Wilkinson bound.
    n = 64
    off = np.arange(n + 1, dtype=np.int32)
    idx = np.arange(n, dtype=np.int32)
    val = np.ones(n, np.float32)
    x = O.xgen_int(n)
    rep = O.rigorous_validate_f32(off, idx, val, x, x.copy())
    assert rep.gpu_overruns == 0 and rep.max_gpu_abs_error == 0.0 and rep.f32_baseline_overruns == 0
    bad = x.copy()
    bad[7] = 1e6
    rep = O.rigorous_validate_f32(off, idx, val, x, bad)
    assert rep.gpu_overruns == 1 and rep.naive_mismatches == 1
    rng = np.random.default_rng(5)
    rows, k = 256, 64
    off = (np.arange(rows + 1) * k).astype(np.int32)
    idx = np.tile(np.arange(k, dtype=np.int32), rows)
    val = (rng.random(rows * k).astype(np.float32) - 0.5) * 1e3
    x = rng.random(k).astype(np.float32)
    y = O.spmv_f32(off, idx, val, x)
    rep = O.rigorous_validate_f32(off, idx, val, x, y)
    assert rep.gpu_overruns <= rep.f32_baseline_overruns + 4 and rep.max_gpu_rel_error < 1e-3
    assert O.count_errors_f32(y, y) == 0


@pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (reference tree absent)")
def test_live_against_reference_build():
    rng = np.random.default_rng(0)
    rows, cols = 2000, 1500
    deg = rng.integers(0, 60, rows)
    off = np.zeros(rows + 1, np.int32)
    off[1:] = np.cumsum(deg)
    nnz = int(off[-1])
    idx = rng.integers(0, cols, nnz).astype(np.int32)
    val = (rng.random(nnz).astype(np.float32) * 2 - 1)
    x = rng.random(cols).astype(np.float32)
    for kind, fn in (("f32", O.spmv_f32), ("f64acc", O.spmv_f64acc_f32), ("l1", O.row_l1_f32)):
        assert np.array_equal(fn(off, idx, val, x), O.ref_spmv_f32(off, idx, val, x, kind=kind)), kind
    for seed in (42, 7, 0xDEADBEEF):
        assert np.array_equal(O.xgen_int(1 << 16, 1, 10, seed), O.ref_xgen_int(1 << 16, 1, 10, seed))
    for R in (2, 3, 4):
        a = O.csr_to_bcsr_f32(R, R, rows, cols, off, idx, val)
        b = O.csr_to_bcsr_f32(R, R, rows, cols, off, idx, val, use_ref=True)
        assert all(np.array_equal(p, q) for p, q in zip(a, b))
    for a, b in ((3.0, 3.01), (1000.0, 1001.5), (0.0, 0.02)):
        assert O.lib().oracle_default_ne_f32(a, b) == O.ref().ref_default_ne_f32(a, b)


# name / rows / cols / nnz / y[0] / sum(y) of the reference's own SpMV battery, as SURVEY.md App. D.3 recorded them from
# the reference's code (unittests/test_spmv_battery.hxx compiled in the survey container)
_D3 = [("identity-16", 16, 16, 16, 1.01729786, 15.1340203), ("banded(0,0)/diag-16", 16, 16, 16, 0.508648932, 110.193568),
       ("banded(1,1)/tridiag-16", 16, 16, 46, 1.09436655, 319.204291), ("banded(3,4)/asym-32", 32, 32, 240, 2.80308247, 3957.75659),
       ("block_diag(4,2)", 8, 8, 16, 1.10488844, 34.570703), ("block_diag(3,3)", 9, 9, 27, 1.85730898, 40.4868238),
       ("skewed(20,50,h=16,l=2)", 20, 50, 54, 15.5284996, 53.468552), ("empty_rows(20,12,0.3,every-4)", 20, 12, 51, 0.0, 48.833135),
       ("random(50,50,0.05)", 50, 50, 133, 6.68259048, 143.51751)]


def test_reference_battery_known_answers():
    """The reference-held fixtures themselves: tests/golden/ref_battery.npz (the restated factories of
    unittests/test_helpers.hxx, mt19937 seeds 7 / 11 / 17 / 23) reproduces every known answer of SURVEY App. D.3, and the
    oracle's reference::spmv restatement reproduces the battery's y bit for bit (same loop, same order)."""
    g = load_golden("ref_battery.npz")
    names = [str(n) for n in g["names"]]
    assert names == [d[0] for d in _D3]
    for k, (name, rows, cols, nnz, y0, total) in enumerate(_D3):
        assert tuple(g[f"{k}.shape"]) == (rows, cols, nnz), name
        y = g[f"{k}.y"]
        assert abs(float(y[0]) - y0) <= 1e-8 * max(1.0, abs(y0)), (name, float(y[0]))
        assert abs(float(y.astype(np.float64).sum()) - total) <= 2e-8 * total, (name, float(y.astype(np.float64).sum()))
        got = O.spmv_f32(g[f"{k}.offsets"], g[f"{k}.indices"], g[f"{k}.values"], g[f"{k}.x"])
        assert np.array_equal(got, y), name
