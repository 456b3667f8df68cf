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


