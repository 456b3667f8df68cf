"""fp64 twins of the format entry points of the C ABI (the reference builds every example as .f32 and .f64,
examples/spmv/CMakeLists.txt:29-50): loops_spmv_{bcsr,coo,csc}_f64 and loops_spmm_merge_path_f64 against the oracle's fp64
restatements, bit-exact on exactly-summable inputs.  (ELL / DIA f64: tests/test_ell_gpu.py, tests/test_dia_gpu.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _matrix(rows=4000, cols=6000, seed=3):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 50, size=rows)
    lens[5] = min(5000, cols)  # a row longer than a merge tile
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) for n in lens]).astype(np.int32)
    val = (rng.integers(1, 9, size=idx.size) / 8.0).astype(np.float64)
    return off, idx, val


def test_coo_csc_f64():
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    off, idx, val = _matrix()
    rows, cols = off.size - 1, 6000
    xh = G.uniform_distribution_int(cols).astype(np.float64)
    want = O.spmv_f64(off, idx, val, xh)
    x = torch.from_numpy(xh).cuda()
    ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(off))
    for tuned in (False, True):
        y = S.coo_spmv(rows, cols, torch.from_numpy(ri).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(val).cuda(), x, tuned=tuned)
        assert y.dtype == torch.float64 and np.array_equal(y.cpu().numpy(), want), ("coo", tuned)
    order = np.lexsort((ri, idx))
    coff = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=cols))]).astype(np.int32)
    for tuned in (False, True):
        y = S.csc_spmv(rows, cols, torch.from_numpy(coff).cuda(), torch.from_numpy(ri[order]).cuda(), torch.from_numpy(val[order]).cuda(),
                       x, tuned=tuned)
        assert y.dtype == torch.float64 and np.array_equal(y.cpu().numpy(), want), ("csc", tuned)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_csc_plan_transposes_once_and_matches_the_oracle(dtype):
    """loops_csc_plan_*: the CSC storage is transposed to CSR on the device once and a SpMV plan runs over the copy.  The copy
    must be the CSR of the same matrix (columns ascending inside a row, the values following them); the product equals the
    oracle's bit for bit, needs no zero-filled y, follows a value refresh, in f32 and f64, with C2-sized skew (x = 8 MB in
    f64, so a measured plan may hold a re-ordered copy) and with empty rows / empty columns."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows, cols = 300_000, 1 << 20
    deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 13)
    deg[::9] = 0
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    val = val.astype(dtype)
    xh = G.uniform_distribution_int(cols).astype(dtype)
    want = (O.spmv_f32(off, idx, val, xh, omp=True) if dtype == np.float32 else O.spmv_f64(off, idx, val, xh))
    ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(off))
    order = np.lexsort((ri, idx))                      # CSC: by column, rows ascending inside a column
    coff = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=cols))]).astype(np.int32)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    x = dev(xh)
    for allow_copy, measure in ((False, False), (True, True)):
        plan = S.CSCPlan(rows, cols, dev(coff), dev(ri[order]), dev(val[order]), allow_copy=allow_copy, measure=measure, repeats=3)
        assert allow_copy or plan.layout == "csr"
        y = torch.full((rows,), -3.0, dtype=x.dtype, device="cuda")
        assert np.array_equal(plan.spmv(x, y).cpu().numpy(), want), (allow_copy, plan.layout)
        plan.refresh_values(dev((3 * val[order]).astype(dtype)))
        assert np.array_equal(plan.spmv(x).cpu().numpy(), 3 * want), (allow_copy, plan.layout)
        plan.close()


@pytest.mark.parametrize("R", [2, 3, 4])
def test_bcsr_f64(R):
    """Register path in fp64 for every compiled block size; the MFMA modes are fp32-only and must say so."""
    from loops_amd import spmv as S, generate as G, _lib
    nbr, per = 512, 12
    boff, bcols, _ = G.uniform_bcsr(nbr, nbr, per, R, R)
    rng = np.random.default_rng(R)
    bvals = (rng.integers(1, 9, size=bcols.size * R * R) / 8.0).astype(np.float64)
    xh = G.uniform_distribution_int(nbr * R).astype(np.float64)
    want = np.zeros(nbr * R, np.float64)
    blocks = bvals.reshape(-1, R, R)
    for br in range(nbr):
        for b in range(boff[br], boff[br + 1]):
            want[br * R:(br + 1) * R] += blocks[b] @ xh[bcols[b] * R:(bcols[b] + 1) * R]
    b = S.BCSR(R, R, nbr * R, nbr * R, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    y = S.bcsr_thread_mapped(b, torch.from_numpy(xh).cuda())
    assert y.dtype == torch.float64 and np.array_equal(y.cpu().numpy(), want)
    if R == 4:
        with pytest.raises(_lib.LoopsError, match="LOOPS_E_CONFIG"):
            S.bcsr_thread_mapped(b, torch.from_numpy(xh).cuda(), mfma=True)


@pytest.mark.parametrize("n", [1, 8, 10, 33])
def test_spmm_plan_f64(n):
    from loops_amd import spmv as S
    from oracle import oracle as O
    off, idx, val = _matrix(rows=3000, cols=2500, seed=8)
    rows, cols = off.size - 1, 2500
    rng = np.random.default_rng(n)
    B = rng.integers(1, 11, size=(cols, n)).astype(np.float64)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    plan = S.MergePathPlan(csr, "256x8")
    got = S.spmm(csr, torch.from_numpy(B).cuda(), plan=plan).cpu().numpy()
    want = np.stack([O.spmv_f64(off, idx, val, np.ascontiguousarray(B[:, j])) for j in range(n)], axis=1)
    assert np.array_equal(got, want)
    assert np.array_equal(S.spmm(csr, torch.from_numpy(B).cuda()).cpu().numpy(), want)  # plan-less f64 entry


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_csc_one_shot_binned_product(dtype):
    """loops_spmv_csc_* mode 1 from 2^20 nonzeros on: the binned product (kernels::launch_csc_binned -- products, one radix pass into bins
    of 4 096 rows, fp64 LDS sums) instead of one atomic per nonzero.  Exactly summable inputs: BIT-EXACT against the CSR product of the
    same matrix; a hub row (a bin shared by several workgroups), empty columns, a row count that is no multiple of the bin, repeated
    calls on one stream (the scratch is reused), and a matrix below the threshold (the atomic kernel) beside it."""
    import scipy.sparse as sp
    from loops_amd import generate as G, spmv as S
    rng = np.random.default_rng(3)
    for rows, cols, nnz_target, hub in ((70_001, 90_000, 1_300_000, 60_000), (300_000, 50_000, 1_100_000, 0), (3000, 4000, 40_000, 0)):
        deg = G.powerlaw_degrees(rows, nnz_target, cap=min(1 << 12, cols)).astype(np.int64)
        if hub:
            deg[12345] = hub                                                   # (two thirds of the columns: its bin is shared by several workgroups)
        off, idx, val = G.csr_from_degrees(deg, cols, 5, 0, True)
        m = sp.csr_matrix((val.astype(np.float64), idx, off), shape=(rows, cols))
        xh = G.uniform_distribution_int(cols)
        want = (m @ xh.astype(np.float64)).astype(dtype)
        c = m.tocsc(); c.sort_indices()
        coff, ridx, cval = c.indptr.astype(np.int32), c.indices.astype(np.int32), c.data.astype(dtype)
        assert (ridx.size >= 1 << 20) == (nnz_target > 1 << 20)
        d = [torch.from_numpy(a).cuda() for a in (coff, ridx, cval, xh.astype(dtype))]
        for _ in range(3):
            y = torch.full((rows,), 9.0, dtype=d[2].dtype, device="cuda")
            S.csc_spmv(rows, cols, d[0], d[1], d[2], d[3], y, tuned=True)
            assert np.array_equal(y.cpu().numpy(), want), (rows, cols, ridx.size)
