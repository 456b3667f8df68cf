"""Panel-binned layout (loops_panel_plan_*, include/loops/kernels/panel_binned.hxx): the device-built layout against its
specification (every nonzero exactly once, in (panel, sub-band, CSR order), 4-item aligned segments, the segment table), and
the SpMV over it against the oracle -- bit for bit on exactly summable inputs, within the measured fp32 bound otherwise,
reproducible from run to run, with the peer fan-out, in f32 and f64, from the battery up to BASELINE C2 / C3 / C5-shard
sizes."""
import numpy as np
import pytest

from conftest import battery, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev(off, idx, val, rows, cols):
    from loops_amd import spmv as S
    return S.CSR.from_numpy(rows, cols, off, idx, val)


def _check_layout(plan, off, idx, val):
    """The layout contract, from the host copies: every nonzero exactly once in (panel, sub-band, CSR) order; groups of 4 never
    straddle segments; dst4 maps every group to where the same items sit in (sub-band, panel, CSR) order; padding is inert.
    Compact plans: col16 bit 15 marks the item that ends a run of equal (row, panel) inside a 256-item window of kernel A,
    the B order holds one slot per run, dst4 bit 31 marks groups that hold padding."""
    v, c16, dst4, r16, perm, bstart = plan.arrays()
    rows, nnz = off.size - 1, idx.size
    W, Hw, P, S = plan.W, plan.Hw, plan.num_panels, plan.num_subbands
    if rows == 0:
        return
    assert plan.padded % 4 == 0 and plan.padded_b % 4 == 0 and (plan.compact or plan.padded_b == plan.padded)
    assert bstart[0] == 0 and bstart[-1] == plan.padded_b and np.all(np.diff(bstart) >= 0) and np.all(bstart % 4 == 0)
    real = perm >= 0
    assert real.sum() == nnz and np.array_equal(np.sort(perm[real]), np.arange(nnz))            # every nonzero exactly once
    assert np.all(v[~real] == 0)                                                                 # padding multiplies to 0
    row_of = np.repeat(np.arange(rows), np.diff(off))
    a = np.flatnonzero(real)                                                                      # A-order positions of the real items
    i = perm[a]
    assert np.array_equal(v[a], val[i])
    p, s = idx[i] // W, row_of[i] // Hw
    assert np.array_equal((c16[a] & (0x7FFF if plan.compact else 0xFFFF)).astype(np.int64), idx[i] - p * W)
    g = p.astype(np.int64) * S + s
    assert np.all(np.diff(g) >= 0) and np.array_equal(np.lexsort((i, g)), np.arange(i.size))     # (panel, sub-band), then CSR order
    grp = a // 4
    same_group = grp[1:] == grp[:-1]
    assert np.all(g[1:][same_group] == g[:-1][same_group])                                        # a group of 4 holds ONE segment
    if not plan.compact:
        b = dst4[grp] + (a % 4)                                                                   # where the item's product goes
        rows_b, seg_s, seg_p, order_key = row_of[i] - s * Hw, s, p, i
    else:
        # run ends: the next real item lies in another segment or row, or the item closes a window of kernel A (64 lanes x 4
        # items; A positions counted from the panel's first item)
        win = plan.a_window
        assert win == 256
        pstart = np.full(P + 1, plan.padded, np.int64)
        np.minimum.at(pstart, p, a)
        win_last = ((a - pstart[p]) % win) == win - 1
        nxt_differs = np.r_[(g[1:] != g[:-1]) | (row_of[i][1:] != row_of[i][:-1]), True]
        ends = nxt_differs | win_last
        assert np.array_equal((c16[a] >> 15).astype(bool), ends)
        assert np.all(c16[~real] == 0)
        assert int(ends.sum()) == plan.runs
        has_pad = np.zeros(plan.padded // 4, bool)
        has_pad[np.flatnonzero(~real) // 4] = True
        assert np.array_equal(dst4[:plan.padded // 4] < 0, has_pad)
        first_slot = dst4[:plan.padded // 4] & 0x7FFFFFFF
        # slot of the run an item belongs to = its group's first slot + the run ends before it inside the group
        ends_before = np.cumsum(ends) - ends                                                      # exclusive, over the real items
        grp_first = np.r_[True, grp[1:] != grp[:-1]]
        base = np.maximum.accumulate(np.where(grp_first, ends_before, 0))
        slot = first_slot[grp] + (ends_before - base)
        b = slot[ends]
        rows_b, seg_s, seg_p, order_key = (row_of[i] - s * Hw)[ends], s[ends], p[ends], i[ends]
        # every item of a run maps to the run's slot: slots are non-decreasing inside a segment and step by one at run ends
        assert np.all(np.diff(slot)[(g[1:] == g[:-1])] == ends[:-1][(g[1:] == g[:-1])])
    assert np.unique(b).size == b.size and b.max(initial=-1) < plan.padded_b
    assert np.array_equal(r16[b].astype(np.int64), rows_b)
    assert np.all((b >= bstart[seg_s]) & (b < bstart[seg_s + 1]))                                 # inside its sub-band's run
    order_b = np.argsort(b, kind="stable")
    gb = seg_s[order_b].astype(np.int64) * P + seg_p[order_b]
    assert np.all(np.diff(gb) >= 0) and np.array_equal(np.lexsort((order_key[order_b], gb)), np.arange(b.size))  # (sub-band, panel), CSR order
    untouched = np.ones(plan.padded_b, bool)
    untouched[b] = False
    assert np.all(r16[untouched] == 0xFFFF)                                                       # padding rows are marked
    # kernel B's work list: the windows of a sub-band tile its run of the B order exactly, <= 256 items each; a window that is
    # not packed lies inside ONE segment (rows sorted); a packed one is made of whole small segments / the short tail of one
    ws, wins, segb = plan.windows()
    pack = 64 if v.dtype == np.float32 else 128
    assert ws[0] == 0 and np.all(np.diff(ws) >= 0) and segb[0] == 0 and segb[-1] == plan.padded_b
    if wins.shape[0] == 0:
        assert plan.padded_b == 0
        return
    wb, wl, wp = wins[:, 0].astype(np.int64), (wins[:, 1] & 0xFFFF).astype(np.int64), wins[:, 1] >> 16
    assert np.all(wl > 0) and np.all(wl <= 256) and np.all(wb % 4 == 0) and np.all(wl % 4 == 0)
    sub = np.repeat(np.arange(S), np.diff(ws))
    assert np.array_equal(wb[ws[:-1][np.diff(ws) > 0]], bstart[:-1][np.diff(ws) > 0])            # a sub-band's first window starts its run
    nxt = np.where(np.r_[sub[1:] != sub[:-1], True], bstart[sub + 1], np.r_[wb[1:], 0])
    assert np.array_equal(wb + wl, nxt)                                                         # contiguous, nothing left out
    assert np.all(bstart[1:][np.diff(ws) == 0] == bstart[:-1][np.diff(ws) == 0])                  # no windows <=> no items
    seg_of = np.searchsorted(segb, wb, side="right") - 1                                          # segment holding the first item
    plain = wp == 0
    assert np.all(wb[plain] + wl[plain] <= segb[seg_of[plain] + 1])                               # one segment
    assert np.all(segb[seg_of[plain] + 1] - segb[seg_of[plain]] > pack)                           # ... a large one
    last_seg = np.searchsorted(segb, wb + wl - 1, side="right") - 1
    inner = np.flatnonzero(~plain)
    for k in inner[:2000]:                                                                        # (spot check: a host loop)
        sizes = np.diff(segb[seg_of[k]:last_seg[k] + 2])
        sizes[0] = segb[seg_of[k] + 1] - wb[k]                                                    # the first piece may be a tail
        assert np.all(sizes <= pack), (k, sizes)


def test_battery_layout_and_product():
    from loops_amd import spmv as S
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        csr = _dev(off, idx, val, r, c)
        for compact in (None, False, True):
            plan = S.PanelBinnedPlan(csr, compact=compact)
            assert compact is None or plan.compact == compact or r == 0
            _check_layout(plan, off, idx, val)
            for tag in ("int", "real"):
                x = torch.from_numpy(g[f"{name}.x_{tag}"]).cuda()
                y = torch.full((r,), 7.0, device="cuda")          # y must not need a zero-fill
                plan.spmv(x, y)
                ref, l1 = g[f"{name}.y_{tag}"], g[f"{name}.l1_{tag}"]
                assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= 1e-6 * l1 + 1e-30), (name, tag, compact)
            plan.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_many_panels_and_subbands_bit_exact(dtype):
    """More columns than one panel and more rows than one sub-band, rows longer than a panel's share, empty rows, a ragged
    tail: bit-exact vs the oracle, the fan-out twin, a value refresh, and two runs give identical bits."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 70_001, 150_001
    deg = G.powerlaw_degrees(rows, 1 << 21, cap=1 << 13)
    deg[::7] = 0
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    xh = G.uniform_distribution_int(cols)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    csr = _dev(off, idx, val.astype(dtype), rows, cols)
    x = torch.from_numpy(xh.astype(dtype)).cuda()
    plan = S.PanelBinnedPlan(csr)
    assert plan.num_panels == -(-cols // plan.W) >= 5 and plan.num_subbands == -(-rows // plan.Hw) > 4
    _check_layout(plan, off, idx, val.astype(dtype))
    y = plan.spmv(x)
    assert np.array_equal(y.cpu().numpy(), ref.astype(dtype))
    assert torch.equal(plan.spmv(x), y)
    peers = [torch.full((rows,), -1.0, dtype=y.dtype, device="cuda") for _ in range(3)]
    y2 = torch.empty_like(y)
    plan.spmv_fanout(x, y2, peers)
    assert torch.equal(y2, y) and all(torch.equal(p, y) for p in peers)
    csr.values.mul_(2.0)
    plan.refresh_values(csr.values)
    assert np.array_equal(plan.spmv(x).cpu().numpy(), 2 * ref.astype(dtype))
    plan.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_small_segments_throughout(dtype):
    """Very short rows over many panels: the mean (panel, sub-band) segment holds far fewer than 64 items -- several segments
    per window of kernel B, so a row may end in several segments of one window and two lanes may target one accumulator in
    the same instruction (the LDS atomics' business).  Bit-exact vs the oracle on exactly summable inputs, identical bits
    from run to run with real values, the fan-out twin, hub rows among the short ones."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 300_007, 1_000_003
    deg = np.full(rows, 2, np.int64)
    deg[::5] = 0
    deg[3::1001] = 700                      # hub rows: ~20 items in every panel of their sub-band
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    nnz = int(off[-1])
    xh = G.uniform_distribution_int(cols)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    csr = _dev(off, idx, val.astype(dtype), rows, cols)
    plan = S.PanelBinnedPlan(csr)
    assert nnz < 64 * plan.num_panels * plan.num_subbands         # small segments throughout
    _check_layout(plan, off, idx, val.astype(dtype))
    x = torch.from_numpy(xh.astype(dtype)).cuda()
    y = plan.spmv(x)
    assert np.array_equal(y.cpu().numpy(), ref.astype(dtype))
    peers = [torch.full((rows,), -1.0, dtype=y.dtype, device="cuda") for _ in range(2)]
    y2 = torch.empty_like(y)
    plan.spmv_fanout(x, y2, peers)
    assert torch.equal(y2, y) and all(torch.equal(p, y) for p in peers)
    xr = torch.from_numpy(G.realistic_x(cols).astype(dtype)).cuda()
    first = plan.spmv(xr).clone()
    for _ in range(5):
        assert torch.equal(plan.spmv(xr), first)
    refr = np.add.reduceat(val.astype(np.float64) * xr.cpu().numpy().astype(np.float64)[idx], np.minimum(off[:-1], nnz - 1).astype(np.int64))
    refr[np.diff(off) == 0] = 0
    l1 = np.add.reduceat(np.abs(val.astype(np.float64) * xr.cpu().numpy().astype(np.float64)[idx]), np.minimum(off[:-1], nnz - 1).astype(np.int64))
    l1[np.diff(off) == 0] = 0
    tol = 1e-6 if dtype == np.float32 else 1e-14
    assert np.all(np.abs(first.cpu().numpy().astype(np.float64) - refr) <= tol * l1 + 1e-300)
    plan.close()


def test_empty_and_degenerate_shapes():
    from loops_amd import spmv as S
    for rows, cols in ((0, 5), (5, 0), (6, 4), (1, 1)):
        off = np.zeros(rows + 1, np.int32)
        csr = _dev(off, np.zeros(0, np.int32), np.zeros(0, np.float32), rows, cols)
        plan = S.PanelBinnedPlan(csr)
        y = torch.full((rows,), 3.0, device="cuda")
        plan.spmv(torch.ones(max(cols, 1), device="cuda"), y)
        assert torch.count_nonzero(y).item() == 0
        plan.close()


def test_real_values_within_the_fp32_bound_and_reproducible():
    from loops_amd import spmv as S, generate as G
    rows = cols = 1 << 18
    deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 13)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, False)       # values U[0.5, 1.5)
    xh = G.realistic_x(cols)
    csr = _dev(off, idx, val, rows, cols)
    plan = S.PanelBinnedPlan(csr)
    x = torch.from_numpy(xh).cuda()
    y = plan.spmv(x)
    ref = np.add.reduceat((val.astype(np.float64) * xh[idx].astype(np.float64)), off[:-1].astype(np.int64))
    ref[np.diff(off) == 0] = 0
    rel = np.abs(y.cpu().numpy().astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)
    # No cancellation here (all terms positive): relative to |y| itself.  The products are fp32 (one rounding each); everything
    # after them -- run sums inside a window, the sub-band's accumulators in LDS -- is fp64, rounded once at the store
    # (panel_reduce_wide): the north star's 1e-6 holds on rows of any length (rows of up to 8 192 nonzeros here; measured
    # 1.1e-7, where round 3's fp32 accumulators reached 2.2e-6 and the reference's sequential CPU loop 5.5e-6).
    assert rel.max() <= 1e-6, rel.max()
    short = np.diff(off) <= 64
    assert rel[short].max() <= 1e-6, rel[short].max()
    for _ in range(5):
        assert torch.equal(plan.spmv(x), y)
    plan.close()


@pytest.mark.parametrize("case", ["c2", "c3_uniform", "c5_shard"])
def test_full_size_configurations_bit_exact(case):
    """BASELINE C2, the C3 stand-in with uniform columns and one C5 shard through the panel-binned plan: equal to the planned
    merge_path_flat product over the unmodified CSR, bit for bit (the latter is pinned against the oracle in test_spmv_gpu.py)."""
    from loops_amd import spmv as S, generate as G
    if case == "c2":
        rows, cols, nnz = 1 << 20, 1 << 20, 1 << 24
    elif case == "c3_uniform":
        rows, cols, nnz = 7_414_866, 7_414_866, 194_109_311
    else:
        rows, cols, nnz = 1 << 21, 1 << 24, 1 << 26
    deg = G.powerlaw_degrees(rows, nnz)
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    csr = _dev(off, idx, val, rows, cols)
    mp = S.MergePathPlan(csr, "512x8")
    want = S.merge_path_flat(csr, x, plan=mp)
    plan = S.PanelBinnedPlan(csr)
    assert plan.padded - nnz <= 3 * plan.num_panels * plan.num_subbands
    got = plan.spmv(x)
    assert torch.equal(got, want), case
    plan.close()


@pytest.mark.parametrize("case", ["c2", "c5_shard"])
def test_full_size_realistic_values_hold_1e6_on_every_row(case):
    """North star: fp32 y within 1e-6 RELATIVE of the f64-accumulated product (util/reference.hxx:146-166 `spmv_f64`, and the
    validator of :278-337) on EVERY row -- BASELINE C2 (rows of up to 2^14 nonzeros) and one rank's shard of C5 with realistic
    values (U[0.5, 1.5) values and x: no cancellation, so the error is relative to |y| itself), through the panel-binned plan
    AND through the SpMV plan a caller holds (loops_spmv_planned_f32; it picks this layout on both inputs).  The twin of
    tests/test_spmv_gpu.py::test_full_size_c2_bit_exact_and_properties for the layout the plan and the N > 1 default run."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    if case == "c2":
        rows, cols, nnz = 1 << 20, 1 << 20, 1 << 24
    else:
        rows, cols, nnz = 1 << 21, 1 << 24, 1 << 26
    deg = G.powerlaw_degrees(rows, nnz)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, False)       # values U[0.5, 1.5)
    xh = G.realistic_x(cols)
    yd = O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64))
    n = np.diff(off.astype(np.int64))
    live = n > 0
    csr = _dev(off, idx, val, rows, cols)
    x = torch.from_numpy(xh).cuda()
    worst = {}
    plan = S.PanelBinnedPlan(csr)
    y = plan.spmv(x)
    for _ in range(3):
        assert torch.equal(plan.spmv(x), y)
    worst["panel_binned"] = (np.abs(y.cpu().numpy().astype(np.float64) - yd)[live] / np.abs(yd[live])).max()
    plan.close()
    held = S.SpmvPlan(csr, allow_copy=True, measure=False)
    yh = held.spmv(x)
    worst["spmv_plan(" + held.info["layout"] + ")"] = (np.abs(yh.cpu().numpy().astype(np.float64) - yd)[live] / np.abs(yd[live])).max()
    held.close()
    print("max relative error vs f64 accumulation:", case, worst, "longest row", int(n.max()))
    assert all(w <= 1e-6 for w in worst.values()), worst
    assert np.all(y.cpu().numpy()[~live] == 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("structure", ["band", "host_blocked"])
def test_compact_layout_on_matrices_with_column_locality(structure, dtype):
    """Column locality (a band around the diagonal; crawl-ordered "hosts"): most of a row's nonzeros share a panel, the plan
    adopts the COMPACT layout by itself (runs <= 0.7 nnz), kernel A pre-sums the runs -- rows longer than a 256-item window
    (run cut at the window), rows spread over several panels, empty rows, a ragged tail.  Bit-exact vs the oracle on exactly
    summable inputs, the same bits as the uncompacted plan, the fan-out twin, a value refresh, identical bits from run to
    run and 1e-6 of the f64 sum with realistic values."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 250_003, 400_009
    deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 12)
    deg[::11] = 0
    if structure == "band":
        off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, 40_000)
    else:
        off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, G.HOST_BLOCKED, hosts=G.host_blocks(cols))
    nnz = int(off[-1])
    xh = G.uniform_distribution_int(cols)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    csr = _dev(off, idx, val.astype(dtype), rows, cols)
    x = torch.from_numpy(xh.astype(dtype)).cuda()
    plan = S.PanelBinnedPlan(csr)
    assert plan.compact and plan.runs <= 0.70 * nnz and plan.padded_b < plan.padded
    assert int(deg.max()) > plan.a_window                               # some run is cut at a window of kernel A
    _check_layout(plan, off, idx, val.astype(dtype))
    y = plan.spmv(x)
    assert np.array_equal(y.cpu().numpy(), ref.astype(dtype))
    flat = S.PanelBinnedPlan(csr, compact=False)
    assert not flat.compact and torch.equal(flat.spmv(x), y)
    flat.close()
    peers = [torch.full((rows,), -1.0, dtype=y.dtype, device="cuda") for _ in range(2)]
    y2 = torch.empty_like(y)
    plan.spmv_fanout(x, y2, peers)
    assert torch.equal(y2, y) and all(torch.equal(p, y) for p in peers)
    # realistic values through a refresh of the held copy
    rng = np.random.default_rng(3)
    val_r = (rng.random(nnz) + 0.5).astype(dtype)
    xr = G.realistic_x(cols).astype(dtype)
    csr.values.copy_(torch.from_numpy(val_r))
    plan.refresh_values(csr.values)
    xd = torch.from_numpy(xr).cuda()
    first = plan.spmv(xd).clone()
    for _ in range(3):
        assert torch.equal(plan.spmv(xd), first)
    live = np.diff(off) > 0
    yd = np.zeros(rows)
    yd[live] = np.add.reduceat(val_r.astype(np.float64) * xr.astype(np.float64)[idx], off[:-1].astype(np.int64)[live])
    rel = np.abs(first.cpu().numpy().astype(np.float64) - yd)[live] / np.abs(yd[live])
    assert rel.max() <= (1e-6 if dtype == np.float32 else 1e-14), rel.max()
    assert np.all(first.cpu().numpy()[~live] == 0)
    plan.close()


def test_out_of_range_column_index_is_refused():
    """A column index outside [0, cols) must not be binned (it would address another panel's keys): LOOPS_E_BADARG."""
    from loops_amd import spmv as S, _lib as L
    off = np.array([0, 2, 3], np.int32)
    for bad in (7, -1):
        idx = np.array([0, bad, 1], np.int32)
        csr = _dev(off, idx, np.ones(3, np.float32), 2, 5)
        with pytest.raises(L.LoopsError, match="BADARG"):
            S.PanelBinnedPlan(csr)


_ROW_BLOCK_SCRIPT = r'''
import numpy as np, torch, sys
from loops_amd import spmv as S, generate as G
from oracle import oracle as O
rows, cols = 6000, 70000
rng = np.random.default_rng(5)
deg = rng.integers(0, 60, rows).astype(np.int64)
deg[1234] = 50000                      # one row longer than a block: a boundary that cannot move
deg[-40:] = 0                          # trailing empty rows belong to the last block
off, idx, val = G.csr_from_degrees(deg, cols, 3)
nnz = int(off[-1])
assert nnz >= 4 * 20000
xh = G.uniform_distribution_int(cols)
ref = O.spmv_f32(off, idx, val, xh)
for dtype in (np.float32, np.float64):
    csr = S.CSR.from_numpy(rows, cols, off, idx, val.astype(dtype))
    x = torch.from_numpy(xh.astype(dtype)).cuda()
    pb = S.PanelBinnedPlan(csr)
    assert pb.row_blocks > 4, pb.row_blocks
    b = pb.row_block_bounds
    assert b[0] == 0 and b[-1] == rows and all(b[i] < b[i + 1] for i in range(len(b) - 1)), b
    y = torch.full((rows,), 7.0, dtype=x.dtype, device="cuda")
    pb.spmv(x, y)
    assert np.array_equal(y.cpu().numpy(), ref.astype(dtype)), dtype
    y.fill_(-1.0)                      # the two kernels of every block, stage by stage
    pb.spmv_stage(0, x, y); pb.spmv_stage(1, x, y)
    assert np.array_equal(y.cpu().numpy(), ref.astype(dtype))
    peers = [torch.full((rows,), -3.0, dtype=x.dtype, device="cuda") for _ in range(2)]
    y.fill_(-1.0)
    pb.spmv_fanout(x, y, peers)
    torch.cuda.synchronize()
    assert all(np.array_equal(p.cpu().numpy(), ref.astype(dtype)) for p in peers) and np.array_equal(y.cpu().numpy(), ref.astype(dtype))
    pb.refresh_values(csr.values * 2)
    assert np.array_equal(pb.spmv(x).cpu().numpy(), 2 * ref.astype(dtype))
    try:
        pb.arrays()
        raise SystemExit("a row-blocked plan has no single set of arrays")
    except Exception as e:
        assert "loops_panel_plan_arrays" in str(e), e
    # a named geometry is one copy, whatever the size
    one = S.PanelBinnedPlan(csr, pb.Hw)
    assert one.row_blocks == 1 and np.array_equal(one.spmv(x).cpu().numpy(), ref.astype(dtype))
    one.close(); pb.close()
    # the held SpMV plan over the same matrix may adopt the blocked copy: same bits either way
    sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=2)
    assert np.array_equal(sp.spmv(x).cpu().numpy(), ref.astype(dtype)), sp.layout
    sp.close()
print("row blocks ok")
'''


def test_row_blocked_plan_equals_one_copy():
    """Matrices of 2^28 nonzeros or more are held as independent panel-binned copies over row blocks (abi_panel.inc); the block
    size is a plan-time knob read once per process (LOOPS_PANEL_BLOCK_NNZ), so the small-scale check runs in its own process:
    product, stage calls, fan-out, value refresh, a row longer than a block, trailing empty rows, f32 / f64."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LOOPS_PANEL_BLOCK_NNZ="20000", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", _ROW_BLOCK_SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "row blocks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
