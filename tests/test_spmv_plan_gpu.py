"""The SpMV plan (loops_spmv_plan_*): tile shape AND layout chosen per matrix -- structural and measured modes, with and without
the permission to hold a re-ordered copy (row-band / panel-binned), value refresh of a held copy, fp64 twin.  Products are
compared with the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_spmv_plan_picks_tile_and_layout():
    """loops_spmv_plan_*: the plan chooses tile shape and layout per matrix.  Structural mode: a band matrix stays on the
    unmodified CSR in 256x8 tiles (self-completing), a matrix whose x is far larger than an L2 is held panel-binned when a copy is allowed and never without that flag.  Measured mode: whatever is chosen, the times of the candidates are
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
    # stays on the CSR takes a phased-gather kernel; round 5: over 256 x 16 tiles from 16 parts of x on)
    assert p.info["layout"] == "csr" and p.info["tile"] == "256x16+phased"
    assert np.array_equal(p.spmv(x).cpu().numpy(), ref)
    p.close()
    p = S.SpmvPlan(csr, allow_copy=True, measure=False)
    assert p.info["layout"] == "panel_binned" and p.info["bands_or_panels"] == cols // 32768      # (4-byte values: panels)
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
        assert ms["csr_256x8"] > 0 and ms["csr_512x8"] > 0 and (ms["row_band"] is None) == (not allow)
        assert (ms["panel_binned"] is None) == (not allow)
        assert allow or p.info["layout"] == "csr"
        if p.info["layout"] == "row_band":
            assert ms["row_band"] < 0.95 * min(ms["csr_256x8"], ms["csr_512x8"])
        if p.info["layout"] == "panel_binned":
            assert ms["panel_binned"] < 0.95 * min(ms["csr_256x8"], ms["csr_512x8"], ms["row_band"])
        assert np.array_equal(p.spmv(x).cpu().numpy(), ref), p.info
        p.close()
    # fp64 twin
    csr64 = S.CSR(csr.rows, csr.cols, csr.offsets, csr.indices, csr.values.double())
    p = S.SpmvPlan(csr64, allow_copy=True, measure=True, repeats=3)
    assert np.array_equal(p.spmv(x.double()).cpu().numpy(), O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64)))
    p.close()


def test_spmv_plan_structural_rule_with_8_byte_values():
    """Without LOOPS_PLAN_MEASURE a copy is chosen by size, whatever the value type: the row-band copy for an x of 2-6 MB under rows
    of >= 8 nonzeros (8-byte values too, since round 6), the panel-binned copy beyond; products equal the oracle's."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    for cols, want in ((1 << 17, "csr"), (1 << 19, "row_band"), (1 << 21, "panel_binned"), (1 << 22, "panel_binned")):  # x = 1 / 4 / 16 / 32 MB
        rows = 1 << 15
        deg = G.powerlaw_degrees(rows, 1 << 20, cap=1 << 12)
        off, idx, val = G.csr_from_degrees(deg, cols, seed=3)
        xh = G.uniform_distribution_int(cols).astype(np.float64)
        csr = S.CSR.from_numpy(rows, cols, off, idx, val.astype(np.float64))
        p = S.SpmvPlan(csr, allow_copy=True, measure=False)
        assert p.info["layout"] == want, (cols, p.info)
        assert np.array_equal(p.spmv(torch.from_numpy(xh).cuda()).cpu().numpy(), O.spmv_f64(off, idx, val.astype(np.float64), xh))
        p.close()
