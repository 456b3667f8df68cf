"""Host-side generators: the reference's x generator (vectorised) and the synthetic workloads."""
import numpy as np

from conftest import load_golden
from loops_amd import generate as G


def test_x_generator_matches_reference_golden():
    g = load_golden("xgen.npz")
    assert np.array_equal(G.uniform_distribution_int(4096, 1, 10, 42), g["x_1_10_42"])
    assert np.array_equal(G.uniform_distribution_int(4096, 0, 1, 7), g["x_0_1_7"])
    assert np.array_equal(G.uniform_distribution_int(4096, -5, 5, 12345), g["x_m5_5_12345"])
    assert np.array_equal(G.uniform_distribution_int(96, 1, 10, 42, start=4000), g["x_1_10_42"][4000:])
    assert list(G.hash32(np.array([0, 1, 2, 41, 12345, 2**32 - 1])).astype(np.uint32)) == list(g["hash"])
    c = load_golden("chesapeake.npz")
    assert np.array_equal(G.uniform_distribution_int(39), c["x"])


def test_powerlaw_generator_properties():
    rows, nnz = 1 << 14, 1 << 18
    deg = G.powerlaw_degrees(rows, nnz, cap=1 << 12)
    assert deg.sum() == nnz and deg.min() >= 1 and deg.max() == 1 << 12
    off, idx, val = G.powerlaw_csr(rows, rows, nnz, degrees=deg)
    assert off[-1] == nnz and idx.min() >= 0 and idx.max() < rows
    r = np.repeat(np.arange(rows), np.diff(off))
    key = (r.astype(np.int64) << 32) | idx
    assert (np.diff(key) > 0).all()  # per-row distinct, ascending
    assert set(np.unique(val * 8)) <= set(range(1, 9))
    # counter-based: a row range generated alone equals the slice of the whole
    o2, i2, v2 = G.powerlaw_csr(rows, rows, nnz, degrees=deg, row_begin=100, row_end=900)
    assert np.array_equal(i2, idx[off[100]:off[900]]) and np.array_equal(v2, val[off[100]:off[900]])
    assert np.array_equal(o2, off[100:901] - off[100])
    # every row sum is exactly representable: f32 and f64 accumulation agree bit-for-bit
    from oracle import oracle as O
    x = G.uniform_distribution_int(rows)
    assert np.array_equal(O.spmv_f32(off, idx, val, x), O.spmv_f64acc_f32(off, idx, val, x))


def test_uniform_bcsr():
    boff, bcols, bvals = G.uniform_bcsr(64, 64, 16)
    assert np.array_equal(np.diff(boff), np.full(64, 16)) and bvals.size == 64 * 16 * 16
    for br in range(64):
        c = bcols[boff[br]:boff[br + 1]]
        assert (np.diff(c) > 0).all()


def test_native_builder_equals_the_numpy_specification():
    """libloops_gen.so (host C++ / OpenMP, used for the 194 M- and 537 M-nonzero configurations) against the numpy
    functions that define the workloads: degrees, CSR rows (uniform / runs / band columns, exact and realistic
    values, a row range generated alone) and the x generator -- bit for bit."""
    G.build_native()
    for rows, nnz, cap in ((1 << 12, 1 << 16, 1 << 10), (100003, 1500007, 1 << 14)):
        d0 = G.powerlaw_degrees(rows, nnz, cap=cap, native=False)
        d1 = G.powerlaw_degrees(rows, nnz, cap=cap, native=True)
        assert np.array_equal(d0, d1) and d1.dtype == d0.dtype
        for window in (None, -1, 64, 4096):
            for exact in (True, False):
                a = G.csr_from_degrees(d0[100:3000], rows, 3, 100, exact, window, native=False)
                b = G.csr_from_degrees(d0[100:3000], rows, 3, 100, exact, window, native=True)
                assert all(np.array_equal(u, v) and u.dtype == v.dtype for u, v in zip(a, b)), (rows, window, exact)
        hosts = G.host_blocks(rows, smin=32)  # host-blocked columns (the third C3 stand-in): hosts of >= 32 ids, power-law sizes
        assert hosts[0] == 0 and hosts[-1] == rows and (np.diff(hosts) > 0).all() and np.diff(hosts)[:-1].min() >= 32
        for exact in (True, False):
            a = G.csr_from_degrees(d0[100:3000], rows, 3, 100, exact, G.HOST_BLOCKED, native=False, hosts=hosts)
            b = G.csr_from_degrees(d0[100:3000], rows, 3, 100, exact, G.HOST_BLOCKED, native=True, hosts=hosts)
            assert all(np.array_equal(u, v) and u.dtype == v.dtype for u, v in zip(a, b)), (rows, "host-blocked", exact)
        ri = np.repeat(np.arange(100, 3000), d0[100:3000])
        j = np.searchsorted(hosts, ri, "right") - 1
        inside = (b[1] >= hosts[j]) & (b[1] < hosts[j + 1])
        assert 0.3 < inside.mean() < 0.9  # most links of short rows stay inside their host; heavy rows go anywhere
        assert np.array_equal(G.uniform_distribution_int(rows + 70000, native=False),
                              G.uniform_distribution_int(rows + 70000, native=True))
        assert np.array_equal(G.uniform_distribution_int(70000, -5, 5, 12345, start=999, native=False),
                              G.uniform_distribution_int(70000, -5, 5, 12345, start=999, native=True))
    # heavy duplicate pressure: 60 distinct columns wanted out of 64 (many redraw rounds)
    deg = np.full(50, 60, np.int64)
    a = G.csr_from_degrees(deg, 64, 5, 0, True, None, native=False)
    b = G.csr_from_degrees(deg, 64, 5, 0, True, None, native=True)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_rmat_generator_native_equals_specification_and_is_scale_free():
    """R-MAT / Kronecker graphs (Graph500 a, b, c = 0.57, 0.19, 0.19; round 5: closer proxies of real graphs than hashed columns):
    the native edge generator equals the numpy specification bit for bit; the CSR keeps every edge, columns sorted inside a row;
    relabelling permutes ids without changing the degree multiset; the degree distribution is heavy-tailed."""
    from loops_amd import generate as G
    r0, c0 = G.rmat_edges(12, 1 << 16, native=False)
    if G.native() is not None:
        r1, c1 = G.rmat_edges(12, 1 << 16, native=True)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)
    assert r0.min() >= 0 and r0.max() < 1 << 12 and c0.min() >= 0 and c0.max() < 1 << 12
    degs = {}
    for relabel in ("none", "random", "degree"):
        off, idx, val = G.rmat_csr(12, 16, relabel=relabel, native=False)
        assert off[0] == 0 and off[-1] == 16 << 12 and idx.size == 16 << 12 and val.min() > 0
        rows = np.repeat(np.arange(1 << 12), np.diff(off))
        assert np.all(np.diff((rows.astype(np.int64) << 32) | idx) >= 0)       # sorted by (row, column), multi-edges kept
        degs[relabel] = np.sort(np.diff(off))
    assert np.array_equal(degs["none"], degs["random"]) and np.array_equal(degs["none"], degs["degree"])
    assert degs["none"][-1] > 50 * 16 and (degs["none"] == 0).mean() > 0.2      # hubs and many isolated vertices
    off, _, _ = G.rmat_csr(12, 16, relabel="degree", native=False)
    d = np.diff(off)
    assert d[:64].sum() > d[-2048:].sum()                                        # hub rows come first after the relabel
