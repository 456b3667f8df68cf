"""CPU checks of the oracle's restatements added for the SpMM and column-blocked paths (the GPU parity
tests compare the kernels against exactly these functions)."""
import numpy as np

from conftest import battery


def _dense(rows, cols, off, idx, val):
    a = np.zeros((rows, cols), np.float64)
    for r in range(rows):
        for k in range(off[r], off[r + 1]):
            a[r, idx[k]] += val[k]
    return a


def test_spmm_oracle_equals_dense_product():
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    for name, (r, c, off, idx, val) in battery().items():
        for n in (1, 5, 16):
            B = rng.integers(1, 11, size=(c, n)).astype(np.float64)
            got = O.spmm(off, idx, val.astype(np.float64), B)
            assert np.allclose(got, _dense(r, c, off, idx, val) @ B, rtol=1e-12, atol=1e-12), (name, n)
            got32 = O.spmm(off, idx, val.astype(np.float32), B.astype(np.float32))
            l1 = np.abs(_dense(r, c, off, idx, np.abs(val))) @ np.abs(B)          # fp32 bound scales with the L1 mass
            assert got32.dtype == np.float32 and np.all(np.abs(got32 - got) <= 8e-6 * l1 + 1e-30), (name, n)


def test_column_blocked_spec_is_a_row_permuted_split():
    """Stacked row k * rows + r holds exactly the nonzeros of row r inside block k, original order;
    summing the stacked rows of a row reproduces the CSR SpMV."""
    from oracle import oracle as O
    for name, (r, c, off, idx, val) in battery().items():
        for K in (1, 2, 5):
            if K > max(c, 1):
                continue
            bounds = np.array([c * k // K for k in range(K + 1)])
            soff, sidx, sval, perm = O.column_blocked(off, idx, val.astype(np.float32), bounds)
            assert soff.size == K * r + 1 and soff[0] == 0 and soff[-1] == idx.size
            assert sorted(perm.tolist()) == list(range(idx.size))
            assert np.array_equal(sidx, np.asarray(idx)[perm]) and np.array_equal(sval, val.astype(np.float32)[perm])
            for k in range(K):
                for row in range(r):
                    seg = slice(soff[k * r + row], soff[k * r + row + 1])
                    assert np.all((sidx[seg] >= bounds[k]) & (sidx[seg] < bounds[k + 1]))
                    assert np.all((perm[seg] >= off[row]) & (perm[seg] < off[row + 1]))
                    assert np.all(np.diff(perm[seg]) > 0)          # original order kept
            x = np.arange(1, c + 1, dtype=np.float64)
            ys = O.spmv_f64(soff, sidx, sval.astype(np.float64), x).reshape(K, r).sum(axis=0)
            ref = O.spmv_f64(off, idx, val.astype(np.float32).astype(np.float64), x)
            assert np.allclose(ys, ref, rtol=1e-10, atol=1e-9)


def test_auto_blocks_rule():
    from oracle import oracle as O
    # long rows: the x-size rule alone (2 MB slices, at most 64)
    assert [O.auto_blocks(c, 1, 1 << 20) for c in (1, 1 << 19, (1 << 19) + 1, 1 << 20, 1 << 22, 1 << 26)] == [1, 1, 2, 2, 8, 64]
    # the row-length cap: half the mean row length, power of two, at least 2
    assert O.auto_blocks(1 << 23, 1 << 20, 1 << 24) == 8      # N = 8 shard of C2: 16 nnz/row
    assert O.auto_blocks(1 << 24, 1 << 21, 1 << 26) == 16     # C5 shard: 32 nnz/row, x = 64 MB
    assert O.auto_blocks(1 << 24, 1 << 20, 1 << 21) == 2      # 2 nnz/row: never more than 2 blocks
    assert O.auto_blocks(1 << 20, 1 << 13, 1 << 17, vbytes=8) == 4
