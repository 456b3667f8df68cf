"""Column-blocked ("stacked") CSR plans through the C ABI (include/loops/kernels/column_blocked.hxx):
the device-built layout must equal the oracle's specification bit for bit (integer work), and the
SpMV over it must equal the plain CSR SpMV (bit-exact for exactly-summable inputs)."""
import numpy as np
import pytest

from conftest import battery, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev(off, idx, val, rows, cols):
    from loops_amd import spmv as S
    return S.CSR.from_numpy(rows, cols, off, idx, val)


def _uniform_bounds(cols, K):
    return np.array([cols * k // K for k in range(K + 1)], np.int64)


@pytest.mark.parametrize("K", [1, 2, 3, 8])
def test_battery_layout_and_spmv(K):
    from loops_amd import spmv as S
    from oracle import oracle as O
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        csr = _dev(off, idx, val.astype(np.float32), r, c)
        plan = S.ColumnBlockedPlan(csr, K)
        k = plan.num_blocks
        assert k == max(1, min(K, c)), (name, k)
        want = O.column_blocked(off, idx, val.astype(np.float32), plan.block_bounds)
        got = plan.arrays()
        for a, b, what in zip(got, want, ("offsets", "indices", "values", "perm")):
            assert np.array_equal(a, b), (name, K, what)
        x = torch.from_numpy(g[f"{name}.x_int"]).cuda()
        y = torch.full((r,), 7.0, device="cuda")
        plan.spmv(x, y)
        ref, l1 = g[f"{name}.y_int"], g[f"{name}.l1_int"]
        assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= 2e-6 * l1 + 1e-30), (name, K)


def test_powerlaw_auto_blocks_bit_exact_and_refresh():
    """x of 16 MB -> 8 automatic blocks; exactly-summable values -> bit-exact vs the plain kernel and
    the oracle; new values through refresh_values."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols, nnz = 1 << 14, 1 << 22, 1 << 18
    off, idx, val = G.csr_from_degrees(G.powerlaw_degrees(rows, nnz, cap=1 << 12), cols, 1, 0, True, None)
    csr = _dev(off, idx, val, rows, cols)
    plan = S.ColumnBlockedPlan(csr)
    assert plan.num_blocks == O.auto_blocks(cols, rows, nnz) == 8
    assert np.array_equal(plan.block_bounds, _uniform_bounds(cols, 8))
    want = O.column_blocked(off, idx, val, plan.block_bounds)
    for a, b in zip(plan.arrays(), want):
        assert np.array_equal(a, b)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    y = plan.spmv(x).cpu().numpy()
    assert np.array_equal(y, O.spmv_f32(off, idx, val, xh))
    assert np.array_equal(y, S.spmv("merge_path_flat", csr, x).cpu().numpy())
    val2 = np.roll(val, 7)
    plan.refresh_values(torch.from_numpy(val2).cuda())
    assert np.array_equal(plan.spmv(x).cpu().numpy(), O.spmv_f32(off, idx, val2, xh))


def test_uneven_bounds_like_row_range_owners():
    """Explicit, uneven column boundaries (what the owners' nnz-balanced row ranges look like),
    including an empty block."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows = cols = 1 << 13
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 17, degrees=G.powerlaw_degrees(rows, 1 << 17, cap=1 << 12))
    csr = _dev(off, idx, val, rows, cols)
    bounds = np.array([0, 100, 100, 3000, 5000, cols], np.int32)
    plan = S.ColumnBlockedPlan(csr, block_bounds=bounds)
    assert plan.num_blocks == 5 and np.array_equal(plan.block_bounds, bounds)
    for a, b in zip(plan.arrays(), O.column_blocked(off, idx, val, bounds)):
        assert np.array_equal(a, b)
    xh = G.uniform_distribution_int(cols)
    assert np.array_equal(plan.spmv(torch.from_numpy(xh).cuda()).cpu().numpy(), O.spmv_f32(off, idx, val, xh))


def test_real_values_and_bad_arguments():
    from loops_amd import spmv as S, generate as G, _lib
    from oracle import oracle as O
    rows = cols = 1 << 12
    off, idx, _ = G.powerlaw_csr(rows, cols, 1 << 16, degrees=G.powerlaw_degrees(rows, 1 << 16, cap=1 << 11))
    rng = np.random.default_rng(2)
    val = rng.uniform(0.5, 1.5, idx.size).astype(np.float32)
    xh = rng.uniform(0.5, 1.5, cols).astype(np.float32)
    csr = _dev(off, idx, val, rows, cols)
    y = S.ColumnBlockedPlan(csr, 4).spmv(torch.from_numpy(xh).cuda()).cpu().numpy()
    ref = O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64))
    l1 = O.row_l1_f32(off, idx, val, xh)
    assert np.all(np.abs(y - ref) <= 2e-6 * l1 + 1e-30)
    with pytest.raises(_lib.LoopsError):
        S.ColumnBlockedPlan(csr, block_bounds=[0, 10, 5, cols])        # not ascending
    with pytest.raises(_lib.LoopsError):
        S.ColumnBlockedPlan(csr, block_bounds=[0, 10, cols - 1])       # does not end at cols
    with pytest.raises(_lib.LoopsError):
        S.ColumnBlockedPlan(csr, 65)                                   # too many blocks


def test_double_precision_plan():
    """f64 plans: x of 2^20 doubles is 8 MB -> 4 automatic blocks; layout and SpMV vs the oracle; the
    value type of the call must match the plan's."""
    from loops_amd import spmv as S, generate as G, _lib
    from oracle import oracle as O
    rows, cols, nnz = 1 << 13, 1 << 20, 1 << 17
    off, idx, val = G.csr_from_degrees(G.powerlaw_degrees(rows, nnz, cap=1 << 12), cols, 1, 0, True, None)
    v64 = val.astype(np.float64)
    csr = _dev(off, idx, v64, rows, cols)
    plan = S.ColumnBlockedPlan(csr)
    assert plan.num_blocks == 4
    for a, b in zip(plan.arrays(), O.column_blocked(off, idx, v64, plan.block_bounds)):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    xh = G.uniform_distribution_int(cols).astype(np.float64)
    y = plan.spmv(torch.from_numpy(xh).cuda()).cpu().numpy()
    assert np.array_equal(y, O.spmv_f64(off, idx, v64, xh))
    plan.refresh_values(torch.from_numpy(np.roll(v64, 3)).cuda())
    assert np.array_equal(plan.spmv(torch.from_numpy(xh).cuda()).cpu().numpy(), O.spmv_f64(off, idx, np.roll(v64, 3), xh))
    with pytest.raises(_lib.LoopsError):   # f32 call on an f64 plan
        _lib.check(_lib.lib().loops_spmv_colblock_f32(plan.handle, 1, 1, None), "loops_spmv_colblock_f32")


def test_spmv_plan_picks_tile_and_layout():
    """loops_spmv_plan_*: the plan chooses tile shape and layout per matrix.  Structural mode: a band matrix stays on the
    unmodified CSR in 256x8 tiles (self-completing), a matrix whose x is far larger than an L2 is held panel-binned (8-byte values below 32
    MB of x: column-blocked) when a copy is allowed and never without that flag.  Measured mode: whatever is chosen, the times of the candidates are
    reported and the product equals the oracle's bit for bit."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    # (a) band matrix, short rows
    rows = cols = 1 << 16
    off, idx, val = G.csr_from_degrees(np.full(rows, 16, np.int64), cols, seed=1, window=64)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    p = S.SpmvPlan(csr, allow_copy=True, measure=False)
    assert p.info["layout"] == "csr" and p.info["tile"] == "256x8" and p.info["measured_ms"]["csr_256x8"] is None
    assert np.array_equal(p.spmv(x).cpu().numpy(), ref)
    p.close()
    assert S.MergePathPlan(csr, "auto").tile == "256x8"
    # (b) power-law rows (longer than a tile), x = 8 M columns = 32 MB
    rows, cols = 1 << 17, 1 << 23
    deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 13)
    off, idx, val = G.csr_from_degrees(deg, cols, seed=1)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    assert S.MergePathPlan(csr, "auto").tile == "512x8"
    p = S.SpmvPlan(csr, allow_copy=False, measure=False)
    # (round 4: hashed columns over an x of 32 MB LOOK scattered -- loops_columns_look_scattered -- so the unmeasured plan that
    # stays on the CSR takes the phased-gather twin of 512 x 8)
    assert p.info["layout"] == "csr" and p.info["tile"] == "512x8+phased"
    assert np.array_equal(p.spmv(x).cpu().numpy(), ref)
    p.close()
    p = S.SpmvPlan(csr, allow_copy=True, measure=False)
    assert p.info["layout"] == "panel_binned" and p.info["column_blocks"] == cols // 32768      # (4-byte values: panels)
    assert np.array_equal(p.spmv(x).cpu().numpy(), ref)
    # new values, same structure: the held copy follows after refresh_values()
    csr.values.mul_(2.0)
    p.refresh_values()
    assert np.array_equal(p.spmv(x).cpu().numpy(), 2.0 * ref)
    csr.values.mul_(0.5)
    p.close()
    p = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=5)   # whichever copy it holds must follow a refresh too
    csr.values.mul_(2.0)
    p.refresh_values()
    assert np.array_equal(p.spmv(x).cpu().numpy(), 2.0 * ref), p.info
    csr.values.mul_(0.5)
    p.close()
    for allow in (False, True):
        p = S.SpmvPlan(csr, allow_copy=allow, measure=True, repeats=5)
        ms = p.info["measured_ms"]
        assert ms["csr_256x8"] > 0 and ms["csr_512x8"] > 0 and (ms["column_blocked"] is None) == (not allow)
        assert (ms["panel_binned"] is None) == (not allow)
        assert allow or p.info["layout"] == "csr"
        if p.info["layout"] == "column_blocked":
            assert ms["column_blocked"] < 0.95 * min(ms["csr_256x8"], ms["csr_512x8"])
        if p.info["layout"] == "panel_binned":
            assert ms["panel_binned"] < 0.95 * min(ms["csr_256x8"], ms["csr_512x8"], ms["column_blocked"])
        assert np.array_equal(p.spmv(x).cpu().numpy(), ref), p.info
        p.close()
    # fp64 twin
    csr64 = S.CSR(csr.rows, csr.cols, csr.offsets, csr.indices, csr.values.double())
    p = S.SpmvPlan(csr64, allow_copy=True, measure=True, repeats=3)
    assert np.array_equal(p.spmv(x.double()).cpu().numpy(), O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64)))
    p.close()


def test_spmv_plan_structural_rule_with_8_byte_values():
    """Without LOOPS_PLAN_MEASURE a copy is chosen by structure: 8-byte values take the panel-binned copy from 32 MB of x, the
    column-blocked one between 6 and 32 MB (where it measured faster: C2 in fp64); products equal the oracle's."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    for cols, want in ((1 << 21, "column_blocked"), (1 << 22, "panel_binned")):       # x = 16 MB / 32 MB
        rows = 1 << 15
        deg = G.powerlaw_degrees(rows, 1 << 20, cap=1 << 12)
        off, idx, val = G.csr_from_degrees(deg, cols, seed=3)
        xh = G.uniform_distribution_int(cols).astype(np.float64)
        csr = S.CSR.from_numpy(rows, cols, off, idx, val.astype(np.float64))
        p = S.SpmvPlan(csr, allow_copy=True, measure=False)
        assert p.info["layout"] == want, (cols, p.info)
        assert np.array_equal(p.spmv(torch.from_numpy(xh).cuda()).cpu().numpy(), O.spmv_f64(off, idx, val.astype(np.float64), xh))
        p.close()
