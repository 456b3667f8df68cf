"""Row-band layout (loops_rowband_plan_*, include/loops/kernels/rowband.hxx): the device-built layout against its numpy
specification (tests/rowband_spec.py: array for array -- items by (band, column, CSR order), one-byte column deltas with padding
slots bridging long gaps, bands padded to steps of 256, the interleave inside a step, hub rows with replicated accumulators,
the chunk list), and the SpMV over it against the oracle
-- bit for bit on exactly summable inputs, within 1e-6 of the f64-accumulated product otherwise, identical bits from run to
run, with the peer fan-out, bands cut into several chunks (second kernel) and uncut, 8 and 16 wavefronts, from the battery up
to BASELINE C2 / C3-band / C5-shard sizes; and the SpMV plan adopting it where it wins."""
import numpy as np
import pytest

from conftest import battery, load_golden
import rowband_spec as spec

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev(off, idx, val, rows, cols):
    from loops_amd import spmv as S
    return S.CSR.from_numpy(rows, cols, off, idx, val)


def _cus():
    import ctypes as C
    from loops_amd import _lib as L
    n = C.c_int()
    L.check(L.lib().loops_device_compute_units(C.byref(n)), "loops_device_compute_units")
    return n.value


def _check_layout(plan, off, idx, val, target_chunks=0):
    """Every array of the device-built plan equals the numpy specification's."""
    rows, cols = off.size - 1, plan.cols
    if rows == 0:
        assert plan.steps == 0 and plan.num_chunks == 0
        return
    v, r16, d8, perm, base, bs, hubs = spec.layout(off, idx, val, rows, cols, plan.H)
    dv, dr16, dd8, dperm, dbase, dchunks, dmulti, dhubs = plan.arrays()
    assert plan.num_bands == -(-rows // plan.H) and plan.steps == base.shape[0]
    assert np.array_equal(dhubs, hubs)
    assert np.array_equal(dperm, perm) and np.array_equal(dr16, r16) and np.array_equal(dd8, d8) and np.array_equal(dv, v)
    assert np.array_equal(dbase, base)
    real = perm >= 0
    assert real.sum() == idx.size and np.array_equal(np.sort(perm[real]), np.arange(idx.size))   # every nonzero exactly once
    assert np.all(v[~real] == 0) and np.all(r16[~real] == plan.H)                                 # inert padding
    assert plan.gap_pads >= int((d8[~real] == 255).sum())                                         # the slots that bridge column gaps (a group's first slot stores delta 0)
    B = plan.num_bands
    target = target_chunks or (B if 2 * B > _cus() else _cus())       # (kernels::rowband_target_chunks)
    ch, mu = spec.chunk_list(bs, target)
    assert np.array_equal(dchunks, ch) and np.array_equal(dmulti, mu)
    assert plan.num_partials == (int(mu[:, 2].sum()) if mu.size else 0)
    # the chunks of a band tile its steps exactly
    for b in range(B):
        mine = ch[ch[:, 0] == b]                        # (list order: by piece number, so a band's chunks appear in piece order)
        assert mine[0, 1] == bs[b] and mine[-1, 2] == bs[b + 1] and np.array_equal(mine[1:, 1], mine[:-1, 2])


def test_battery_layout_and_product():
    from loops_amd import spmv as S
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        csr = _dev(off, idx, val, r, c)
        for band_rows, target in ((0, 0), (64, 0), (64, 7)):
            plan = S.RowBandPlan(csr, band_rows, target)
            _check_layout(plan, off, idx, val, target)
            for waves in (8, 16):
                plan.set_waves(waves)
                for tag in ("int", "real"):
                    x = torch.from_numpy(g[f"{name}.x_{tag}"]).cuda()
                    y = torch.full((r,), 7.0, device="cuda")          # y must not need a zero-fill
                    plan.spmv(x, y)
                    ref, l1 = g[f"{name}.y_{tag}"], g[f"{name}.l1_{tag}"]
                    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= 1e-6 * l1 + 1e-30), (name, tag, band_rows, target, waves)
            plan.close()


@pytest.mark.parametrize("band_rows,target", [(0, 0), (16384, 0), (8192, 700), (2048, 0), (256, 1000)])
def test_many_bands_bit_exact(band_rows, target):
    """More rows than one band, columns beyond 2^16, power-law rows with hubs (replicated accumulators), empty
    rows, a ragged tail; bands uncut and cut into several chunks (partial vectors + the second kernel): bit-exact vs the oracle,
    the numpy specification of the product, the fan-out twin, a value refresh, and two runs give identical bits."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 70_001, 150_001
    deg = G.powerlaw_degrees(rows, 1 << 21, cap=1 << 13)
    deg[::7] = 0
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    xh = G.uniform_distribution_int(cols)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(xh).cuda()
    plan = S.RowBandPlan(csr, band_rows, target)
    assert plan.num_bands == -(-rows // plan.H)
    _check_layout(plan, off, idx, val, target)
    v, r16, d8, perm, base, bs, hubs = spec.layout(off, idx, val, rows, cols, plan.H)
    assert hubs[:, 0].max() > 0                                          # some band has hub rows
    assert np.array_equal(spec.product(v, r16, d8, base, bs, hubs, plan.H, rows, xh).astype(np.float32), ref)
    if target:
        assert plan.num_partials > 0
    for waves in (8, 16):
        plan.set_waves(waves)
        y = plan.spmv(x)
        assert np.array_equal(y.cpu().numpy(), ref), waves
        assert torch.equal(plan.spmv(x), y)
    peers = [torch.full((rows,), -1.0, device="cuda") for _ in range(3)]
    y2 = torch.empty_like(y)
    plan.spmv_fanout(x, y2, peers)
    assert torch.equal(y2, y) and all(torch.equal(p, y) for p in peers)
    csr.values.mul_(2.0)
    plan.refresh_values(csr.values)
    assert np.array_equal(plan.spmv(x).cpu().numpy(), 2 * ref)
    ms8, ms16 = plan.tune(3)
    assert ms8 > 0 and ms16 > 0 and plan.waves in (8, 16)
    assert np.array_equal(plan.spmv(x).cpu().numpy(), 2 * ref)
    plan.close()


def test_short_rows_with_long_column_gaps():
    """Very short rows over a million columns: the mean column gap inside a band is far beyond 255, so padding slots that bridge
    the gaps outnumber the nonzeros (dump-word adds of x values that are read but never used), hub rows among the short ones.  Bit-exact on exactly summable inputs; identical bits from run to
    run and 1e-6 with real values."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 300_007, 1_000_003
    deg = np.full(rows, 2, np.int64)
    deg[::5] = 0
    deg[3::1001] = 700
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    nnz = int(off[-1])
    xh = G.uniform_distribution_int(cols)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    csr = _dev(off, idx, val, rows, cols)
    plan = S.RowBandPlan(csr)
    assert plan.gap_pads > 0.05 * nnz and plan.padded >= nnz + plan.gap_pads        # (mean column gap of a band ~ 90: one in ten beyond 255)
    _check_layout(plan, off, idx, val)
    x = torch.from_numpy(xh).cuda()
    assert np.array_equal(plan.spmv(x).cpu().numpy(), ref)
    xr = torch.from_numpy(G.realistic_x(cols)).cuda()
    first = plan.spmv(xr).clone()
    for _ in range(5):
        assert torch.equal(plan.spmv(xr), first)
    prod = val.astype(np.float64) * xr.cpu().numpy().astype(np.float64)[idx]
    at = np.minimum(off[:-1], nnz - 1).astype(np.int64)
    refr, l1 = np.add.reduceat(prod, at), np.add.reduceat(np.abs(prod), at)
    refr[np.diff(off) == 0] = 0
    l1[np.diff(off) == 0] = 0
    assert np.all(np.abs(first.cpu().numpy().astype(np.float64) - refr) <= 1e-6 * l1 + 1e-300)
    plan.close()


def test_empty_and_degenerate_shapes():
    from loops_amd import spmv as S
    for rows, cols in ((0, 5), (5, 0), (6, 4), (1, 1)):
        off = np.zeros(rows + 1, np.int32)
        csr = _dev(off, np.zeros(0, np.int32), np.zeros(0, np.float32), rows, cols)
        plan = S.RowBandPlan(csr)
        y = torch.full((rows,), 3.0, device="cuda")
        plan.spmv(torch.ones(max(cols, 1), device="cuda"), y)
        assert torch.count_nonzero(y).item() == 0
        plan.close()


def test_bad_arguments_are_refused():
    """A column index outside [0, cols) must not be binned; band heights that are not a power of two in [64, 16384]."""
    from loops_amd import spmv as S, _lib as L
    off = np.array([0, 2, 3], np.int32)
    for bad in (7, -1):
        csr = _dev(off, np.array([0, bad, 1], np.int32), np.ones(3, np.float32), 2, 5)
        with pytest.raises(L.LoopsError, match="BADARG"):
            S.RowBandPlan(csr)
    csr = _dev(off, np.array([0, 1, 1], np.int32), np.ones(3, np.float32), 2, 5)
    for h in (32, 100, 32768):
        with pytest.raises(L.LoopsError, match="BADARG"):
            S.RowBandPlan(csr, h)


def test_real_values_within_the_fp32_bound_and_reproducible():
    from loops_amd import spmv as S, generate as G
    rows = cols = 1 << 18
    deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 13)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, False)       # values U[0.5, 1.5)
    xh = G.realistic_x(cols)
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(xh).cuda()
    ref = np.add.reduceat((val.astype(np.float64) * xh[idx].astype(np.float64)), off[:-1].astype(np.int64))
    ref[np.diff(off) == 0] = 0
    for target in (0, 1024):                                            # uncut / cut bands (fp32 partial vectors: one more rounding)
        plan = S.RowBandPlan(csr, 0, target)
        y = plan.spmv(x)
        rel = np.abs(y.cpu().numpy().astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)
        # no cancellation here (all terms positive): relative to |y| itself.  Products are fp32 (one rounding each), sums fp64.
        assert rel.max() <= 1e-6, (target, rel.max())
        for _ in range(5):
            assert torch.equal(plan.spmv(x), y)
        plan.close()


@pytest.mark.parametrize("case", ["c2", "c3_band65536", "c5_shard"])
def test_full_size_configurations_bit_exact(case):
    """BASELINE C2, the C3 stand-in with a 65 536-wide band and one C5 shard through the row-band plan: equal to the planned
    merge_path_flat product over the unmodified CSR, bit for bit (the latter is pinned against the oracle in test_spmv_gpu.py)."""
    from loops_amd import spmv as S, generate as G
    window = None
    if case == "c2":
        rows, cols, nnz = 1 << 20, 1 << 20, 1 << 24
    elif case == "c3_band65536":
        rows, cols, nnz, window = 7_414_866, 7_414_866, 194_109_311, 65536
    else:
        rows, cols, nnz = 1 << 21, 1 << 24, 1 << 26
    deg = G.powerlaw_degrees(rows, nnz)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    csr = _dev(off, idx, val, rows, cols)
    mp = S.MergePathPlan(csr, "512x8")
    want = S.merge_path_flat(csr, x, plan=mp)
    plan = S.RowBandPlan(csr)
    assert plan.H == 16384 and plan.padded - nnz - plan.gap_pads <= 255 * plan.num_bands
    for waves in (8, 16):
        plan.set_waves(waves)
        got = plan.spmv(x)
        assert torch.equal(got, want), (case, waves)
    plan.close()


def test_full_size_c2_realistic_values_hold_1e6_on_every_row():
    """North star: fp32 y within 1e-6 RELATIVE of the f64-accumulated product (util/reference.hxx:146-166 `spmv_f64`, the validator
    of :278-337) on EVERY row of BASELINE C2 (rows of up to 2^14 nonzeros, realistic values: no cancellation), through the
    row-band plan AND through the SpMV plan a caller holds (loops_spmv_planned_f32 picks this layout on C2)."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols, nnz = 1 << 20, 1 << 20, 1 << 24
    deg = G.powerlaw_degrees(rows, nnz)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, False)       # values U[0.5, 1.5)
    xh = G.realistic_x(cols)
    yd = O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64))
    n = np.diff(off.astype(np.int64))
    live = n > 0
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(xh).cuda()
    worst = {}
    plan = S.RowBandPlan(csr)
    y = plan.spmv(x)
    for _ in range(3):
        assert torch.equal(plan.spmv(x), y)
    worst["row_band"] = (np.abs(y.cpu().numpy().astype(np.float64) - yd)[live] / np.abs(yd[live])).max()
    plan.close()
    held = S.SpmvPlan(csr, allow_copy=True, measure=True)
    assert held.layout == "row_band", held.info
    yh = held.spmv(x)
    worst["spmv_plan(" + held.layout + ")"] = (np.abs(yh.cpu().numpy().astype(np.float64) - yd)[live] / np.abs(yd[live])).max()
    held.close()
    print("max relative error vs f64 accumulation:", worst, "longest row", int(n.max()))
    assert all(w <= 1e-6 for w in worst.values()), worst
    assert np.all(y.cpu().numpy()[~live] == 0)


def test_spmv_plan_adopts_the_row_band_copy_where_it_wins_and_refreshes_it():
    """A C2-like matrix at a quarter of the size (x = 1 MB... 2 MB): with MEASURE + ALLOW_COPY the plan times the row-band copy
    (ms reported), whatever it keeps computes the CSR product's bits, and a value refresh reaches the held copy."""
    from loops_amd import spmv as S, generate as G
    rows = cols = 1 << 19
    deg = G.powerlaw_degrees(rows, 1 << 23, cap=1 << 13)
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    mp = S.MergePathPlan(csr, "512x8")
    want = S.merge_path_flat(csr, x, plan=mp)
    held = S.SpmvPlan(csr, allow_copy=True, measure=True)
    assert held.measured_ms["row_band"] is not None and held.measured_ms["row_band"] > 0
    assert torch.equal(held.spmv(x), want), held.info
    structural = S.SpmvPlan(csr, allow_copy=True, measure=False)
    assert structural.layout == "row_band" and torch.equal(structural.spmv(x), want)   # x = 2 MB, mean row 16
    csr.values.mul_(2.0)
    for p in (held, structural):
        p.refresh_values()
        assert torch.equal(p.spmv(x), 2 * want)
        p.close()


# ----------------------------------------------------------------------------------------------------------------------
# 8-byte values (the reference builds every example as .f32 and .f64, examples/spmv/CMakeLists.txt:29-50): the same layout
# arrays (the record per slot does not depend on the value type), fp64 products and sums, fp64 partial vectors.
@pytest.mark.parametrize("band_rows,target", [(0, 0), (8192, 700), (256, 1000)])
def test_f64_many_bands_bit_exact(band_rows, target):
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 70_001, 150_001
    deg = G.powerlaw_degrees(rows, 1 << 21, cap=1 << 13)
    deg[::7] = 0
    off, idx, val32 = G.csr_from_degrees(deg, cols, 1)
    val = val32.astype(np.float64)
    xh = G.uniform_distribution_int(cols).astype(np.float64)
    ref = O.spmv_f64(off, idx, val, xh)
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(xh).cuda()
    plan = S.RowBandPlan(csr, band_rows, target)
    assert plan.dtype == torch.float64 and plan.num_bands == -(-rows // plan.H)
    _check_layout(plan, off, idx, val, target)
    if target:
        assert plan.num_partials > 0
    for waves in (8, 16):
        plan.set_waves(waves)
        y = torch.full((rows,), 7.0, dtype=torch.float64, device="cuda")
        plan.spmv(x, y)
        assert np.array_equal(y.cpu().numpy(), ref), waves
    csr.values.mul_(2.0)
    plan.refresh_values(csr.values)
    assert np.array_equal(plan.spmv(x).cpu().numpy(), 2 * ref)
    ms8, ms16 = plan.tune(3)
    assert ms8 > 0 and ms16 > 0
    plan.close()


def test_f64_battery_and_real_values():
    """The battery in fp64 (golden y of the f64 reference build) and realistic values: within 1e-13 of the exact product."""
    from loops_amd import spmv as S
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        csr = _dev(off, idx, val.astype(np.float64), r, c)
        plan = S.RowBandPlan(csr, 64, 7)
        for tag in ("int", "real"):
            xh = g[f"{name}.x_{tag}"].astype(np.float64)
            y = plan.spmv(torch.from_numpy(xh).cuda()).cpu().numpy()
            prod = val.astype(np.float64) * xh[idx]
            want = np.zeros(r)
            np.add.at(want, np.repeat(np.arange(r), np.diff(off)), prod)
            l1 = np.zeros(r)
            np.add.at(l1, np.repeat(np.arange(r), np.diff(off)), np.abs(prod))
            assert np.all(np.abs(y - want) <= 1e-13 * l1 + 1e-300), (name, tag)
        plan.close()


def test_f64_spmv_plan_can_hold_the_row_band_copy():
    """An f64 SpMV plan may now choose the row-band copy (unmeasured: by size; LOOPS_PLAN_DETERMINISTIC rules it out)."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows = cols = 1 << 18                                                  # x = 2 MB in fp64
    deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 12)
    off, idx, val32 = G.csr_from_degrees(deg, cols, 1)
    val = val32.astype(np.float64)
    xh = G.uniform_distribution_int(cols).astype(np.float64)
    ref = O.spmv_f64(off, idx, val, xh)
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(xh).cuda()
    plan = S.SpmvPlan(csr, measure=False, allow_copy=True)
    assert plan.layout == "row_band", plan.layout
    assert np.array_equal(plan.spmv(x).cpu().numpy(), ref)
    det = S.SpmvPlan(csr, measure=False, allow_copy=True, deterministic=True)
    assert det.layout == "csr", det.layout
    assert np.array_equal(det.spmv(x).cpu().numpy(), ref)


def test_band_count_just_under_the_cu_count_is_left_uncut():
    """More than half as many bands as compute units: one chunk per band (cutting ONE band of 255 on 256 CUs would buy a partial-vector
    round trip and a second launch for nothing); half as many or fewer: one round of equal chunks.  The layout still equals the
    specification and the product the oracle's."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    cus = _cus()
    for bands in (cus - 1, cus // 2 + 1, cus // 2, 3):
        rows = bands * 64 - 5
        deg = np.random.default_rng(bands).integers(0, 9, size=rows)
        off, idx, val = G.csr_from_degrees(deg.astype(np.int64), 5000, seed=bands)
        csr = _dev(off, idx, val, rows, 5000)
        plan = S.RowBandPlan(csr, 64, 0)
        assert plan.num_bands == bands
        assert (plan.num_chunks == bands and plan.num_partials == 0) if 2 * bands > cus else plan.num_chunks >= min(cus, plan.steps)
        _check_layout(plan, off, idx, val, 0)
        xh = G.uniform_distribution_int(5000)
        assert np.array_equal(plan.spmv(torch.from_numpy(xh).cuda()).cpu().numpy(), O.spmv_f32(off, idx, val, xh))
        plan.close()
