"""Randomised shape sweep over every tuned entry point of the C ABI against the CPU oracle: many small
matrices with empty rows, single huge rows, rows spanning merge tiles, more columns than rows and vice
versa.  Inputs are exactly summable (values k/8, integer x) so every comparison is BIT-EXACT whatever
the summation order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _random_csr(rng, rows, cols, kind):
    if kind == "empty":
        lens = np.zeros(rows, np.int64)
    elif kind == "uniform":
        lens = rng.integers(0, min(cols, 12) + 1, size=rows)
    elif kind == "skewed":        # a few rows hold most nonzeros (rows longer than a 2048-item merge tile)
        lens = rng.integers(0, 4, size=rows)
        for r in rng.choice(rows, size=min(rows, 3), replace=False):
            lens[r] = min(cols, int(rng.integers(1500, 6000)))
    else:                         # "ragged": geometric tail
        lens = np.minimum(rng.geometric(0.08, size=rows) - 1, cols)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.sort(rng.choice(cols, size=int(n), replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = (rng.integers(-8, 9, size=idx.size) / 8.0).astype(np.float32)
    return off, idx, val


import os
# LOOPS_FUZZ_SEEDS=N widens the sweep (default 20 seeds x 4 kinds = 80 matrices; a 200-seed run takes a few GPU-minutes)
CASES = [(seed, kind) for seed in range(int(os.environ.get("LOOPS_FUZZ_SEEDS", "20"))) for kind in ("empty", "uniform", "skewed", "ragged")]


@pytest.mark.parametrize("seed,kind", CASES)
def test_every_tuned_path_matches_the_oracle(seed, kind):
    from loops_amd import spmv as S
    from oracle import oracle as O
    rng = np.random.default_rng(1000 * seed + len(kind))
    rows = int(rng.integers(1, 3000))
    cols = int(rng.integers(1, 7000)) if kind != "skewed" else int(rng.integers(6000, 9000))
    off, idx, val = _random_csr(rng, rows, cols, kind)
    xh = rng.integers(1, 11, size=cols).astype(np.float32)
    want = O.spmv_f32(off, idx, val, xh)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(xh).cuda()
    tag = (seed, kind, rows, cols, idx.size)
    # CSR schedules
    for sch in ("merge_path_flat", "work_oriented", "group_mapped", "thread_mapped", "original", "flat_partitioned"):
        y = torch.full((rows,), 5.0, device="cuda")
        assert np.array_equal(S.spmv(sch, csr, x, y).cpu().numpy(), want), (sch,) + tag
    # held plan (self-completing single-kernel path when every tile head is short, two kernels otherwise)
    for tile in ("256x8", "128x7"):
        for variant in (0, 4):   # 0 = bit-mask split (the default), 4 = per-thread halving search
            y = torch.full((rows,), 5.0, device="cuda")
            S.merge_path_flat(csr, x, y, plan=S.MergePathPlan(csr, tile), variant=variant)
            assert np.array_equal(y.cpu().numpy(), want), ("planned", tile, variant) + tag
    # row-band copy: automatic and smallest bands, uncut and cut into chunks, both kernel shapes
    for hb, target, waves in ((0, 0, 8), (64, 0, 16), (64, 9, 8)):
        rb = S.RowBandPlan(csr, hb, target)
        rb.set_waves(waves)
        y = torch.full((rows,), 5.0, device="cuda")
        assert np.array_equal(rb.spmv(x, y).cpu().numpy(), want), ("row_band", hb, target, waves) + tag
        rb.close()
    # panel-binned copy (automatic and smallest sub-bands) and the measured SpMV plan that may pick it
    # (both forms of the B order: one slot per nonzero, and -- compact -- one per run of equal (row, panel) pre-summed by kernel A;
    # with <= 9000 columns everything is one panel, so a row is one run cut only at kernel A's 256-item windows)
    for hw in (0, 64):
        for compact in (None, False, True):
            pb = S.PanelBinnedPlan(csr, hw, compact=compact)
            y = torch.full((rows,), 5.0, device="cuda")
            assert np.array_equal(pb.spmv(x, y).cpu().numpy(), want), ("panel", hw, compact, pb.compact) + tag
            pb.close()
    sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=1)
    assert np.array_equal(sp.spmv(x).cpu().numpy(), want), ("spmv_plan", sp.info["layout"]) + tag
    sp.close()
    # COO (sorted and shuffled) and ELL
    ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(off))
    for perm in (np.arange(idx.size), rng.permutation(idx.size)):
        y = S.coo_spmv(rows, cols, torch.from_numpy(ri[perm]).cuda(), torch.from_numpy(idx[perm]).cuda(),
                       torch.from_numpy(val[perm]).cuda(), x)
        assert np.array_equal(y.cpu().numpy(), want), ("coo",) + tag
    pitch = int(np.diff(off).max()) if rows else 0
    if rows * max(pitch, 1) < 4_000_000:
        ind = np.full((rows, max(pitch, 1)), -1, np.int32)[:, :pitch]
        ev = np.zeros((rows, pitch), np.float32)
        for r in range(rows):
            n = off[r + 1] - off[r]
            ind[r, :n] = idx[off[r]:off[r + 1]]
            ev[r, :n] = val[off[r]:off[r + 1]]
        ind_d, ev_d = torch.from_numpy(np.ascontiguousarray(ind)).cuda(), torch.from_numpy(np.ascontiguousarray(ev)).cuda()
        for mode in (True, "merge_path"):   # row-split kernel; merge_path_flat over the ELL cells on the fused engine
            # rows == 0 returns before any launch; pitch == 0 (every row empty) runs the merge path over row ends only:
            # null index / value arrays, natoms = 0 in every tile, y = 0
            y = torch.full((rows,), 5.0, device="cuda")
            S.ell_spmv(rows, cols, pitch, ind_d, ev_d, x, y, tuned=mode)
            assert np.array_equal(y.cpu().numpy(), want), ("ell", mode) + tag
    # DIA (small matrices only: one stored diagonal per distinct col - row), both kernels, f32 and f64
    if 0 < rows <= 300 and cols <= 300 and idx.size:
        d = idx.astype(np.int64) - ri.astype(np.int64)
        diags = np.unique(d)
        cells = np.zeros((diags.size, rows), np.float32)
        cells[np.searchsorted(diags, d), ri] = val
        for tuned in (False, True):
            y = S.dia_spmv(rows, cols, torch.from_numpy(diags.astype(np.int32)).cuda(), torch.from_numpy(cells).cuda(), x, tuned=tuned)
            assert np.array_equal(y.cpu().numpy(), want), ("dia", tuned) + tag
            y = S.dia_spmv(rows, cols, torch.from_numpy(diags.astype(np.int32)).cuda(), torch.from_numpy(cells.astype(np.float64)).cuda(),
                           x.double(), tuned=tuned)
            assert np.array_equal(y.cpu().numpy(), want.astype(np.float64)), ("dia f64", tuned) + tag
    # CSC (built on the host from the CSR), both kernels
    order = np.lexsort((ri, idx))                      # by column, rows ascending inside a column
    coff = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=cols))]).astype(np.int32)
    for tuned in (False, True):
        y = S.csc_spmv(rows, cols, torch.from_numpy(coff).cuda(), torch.from_numpy(ri[order]).cuda(),
                       torch.from_numpy(val[order]).cuda(), x, tuned=tuned)
        assert np.array_equal(y.cpu().numpy(), want), ("csc", tuned) + tag
    for measure in (False, True):   # the held CSC plan (transposed once on the device), incl. a value refresh
        cp = S.CSCPlan(rows, cols, torch.from_numpy(coff).cuda(), torch.from_numpy(ri[order]).cuda(), torch.from_numpy(val[order]).cuda(),
                       allow_copy=True, measure=measure, repeats=2)
        yp = torch.full((rows,), 5.0, device="cuda")
        assert np.array_equal(cp.spmv(x, yp).cpu().numpy(), want), ("csc plan", measure) + tag
        cp.refresh_values(torch.from_numpy(2 * val[order]).cuda())
        assert np.array_equal(cp.spmv(x).cpu().numpy(), 2 * want), ("csc plan refresh", measure) + tag
        cp.close()
    # COO plan: shuffled triplets sorted once on the device
    shuffle = rng.permutation(idx.size)
    kp = S.COOPlan(rows, cols, torch.from_numpy(ri[shuffle]).cuda(), torch.from_numpy(idx[shuffle]).cuda(), torch.from_numpy(val[shuffle]).cuda(),
                   allow_copy=False, measure=False)
    assert np.array_equal(kp.spmv(x).cpu().numpy(), want), ("coo plan",) + tag
    kp.close()
    # SpMM, a few widths of B
    for n in (1, 6, 16, 40):
        B = rng.integers(1, 11, size=(cols, n)).astype(np.float32)
        got = S.spmm(csr, torch.from_numpy(B).cuda()).cpu().numpy()
        assert np.array_equal(got, O.spmm(off, idx, val, B)), ("spmm", n) + tag
