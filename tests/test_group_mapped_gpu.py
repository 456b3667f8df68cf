"""group_mapped with heavy groups shared out (kernels/group_mapped_spmv.hxx; reference schedule/group_mapped.hxx:104-192 for the
ownership rule, algorithms/spmv/group_mapped.cuh:27-61 for the kernel it replaces): groups of more than 24 merge tiles are
published and swept by a second launch (a workgroup per claim of 2 tiles), carry-outs of the claims are added by a fix-up launch;
the one-shot entry remembers per matrix whether anything was published and then launches the one-kernel form.  Exactly summable
inputs -> BIT-EXACT against the oracle whatever the order of the claims; realistic values within 1e-6 of the f64 product."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TILE = 2048          # 256 x 8
CLAIM = 2 * TILE     # kernels::group_claim_tiles


def _csr(degrees, cols, seed, exact=True):
    from loops_amd import generate as G
    return G.csr_from_degrees(np.asarray(degrees, np.int64), cols, seed, 0, exact)


def _run(off, idx, val, rows, cols, x, sched="group_mapped"):
    from loops_amd import spmv as S
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    y = torch.full((rows,), -3.0, dtype=torch.from_numpy(val).dtype, device="cuda")
    S.spmv(sched, csr, torch.from_numpy(x).cuda(), y)
    return y.cpu().numpy()


def _degree_cases():
    rng = np.random.default_rng(4)
    cases = {}
    d = rng.integers(0, 9, size=3000); d[0] = 60_000                                  # one hub in the first group: 30 tiles, 15 claims
    cases["hub_first_row"] = d
    d = rng.integers(0, 9, size=3000); d[255] = 60_000; d[256] = 60_000               # hubs at a group boundary: two heavy groups
    cases["hubs_across_groups"] = d
    d = rng.integers(0, 5, size=1000); d[700:760] = 3_000                             # many medium rows: a heavy group without one long row
    cases["dense_group"] = d
    d = rng.integers(0, 9, size=2900); d[-1] = 50_000; d[-300] = 30_000               # heavy LAST (partial) group, hub is the last row
    cases["heavy_partial_last_group"] = d
    d = np.zeros(600, np.int64); d[10] = 100_000; d[300] = 17                         # empty rows around a hub (row spans > one claim)
    cases["hub_among_empty_rows"] = d
    d = np.full(256 * 6, 200, np.int64)                                               # every group 26 tiles: all groups heavy, no long rows
    cases["all_groups_heavy"] = d
    d = rng.integers(0, 9, size=5000); d[::256] = 60_000                              # a hub at the start of each of 20 groups
    cases["hub_per_group"] = d
    d = rng.integers(0, 9, size=2000); d[5] = CLAIM * 9; d[6] = CLAIM - 1; d[7] = 1; d[8] = TILE  # rows ending exactly on claim / tile edges
    cases["rows_on_claim_edges"] = d
    return cases


@pytest.mark.parametrize("name", sorted(_degree_cases()))
def test_shared_out_groups_bit_exact(name):
    from loops_amd import generate as G
    from oracle import oracle as O
    deg = _degree_cases()[name]
    rows, cols = deg.size, 1 << 17
    off, idx, val = _csr(deg, cols, seed=len(name))
    x = G.uniform_distribution_int(cols)
    want = O.spmv_f32(off, idx, val, x)
    for _ in range(3):                                                                # (the scratch must come back clean: repeat)
        assert np.array_equal(_run(off, idx, val, rows, cols, x), want), name
    assert np.array_equal(_run(off, idx, val, rows, cols, x, "merge_path_flat"), want)


def test_problem_sizes_alternate_on_one_stream():
    """The per-stream scratch is laid out per problem size: alternating sizes (with and without heavy groups) must not see each
    other's carry-outs as published groups."""
    from loops_amd import generate as G
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    mats = []
    for rows, hub in ((4000, 90_000), (700, 0), (9000, 70_000), (300, 20_000), (4000, 90_000)):
        d = rng.integers(0, 12, size=rows)
        if hub:
            d[rows // 3] = hub
        off, idx, val = _csr(d, 1 << 17, seed=rows)
        x = G.uniform_distribution_int(1 << 17, seed=rows)
        mats.append((off, idx, val, rows, x, O.spmv_f32(off, idx, val, x)))
    for _ in range(2):
        for off, idx, val, rows, x, want in mats:
            assert np.array_equal(_run(off, idx, val, rows, 1 << 17, x), want), rows


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_realistic_values_within_tolerance(dtype):
    """Values and x in [0.5, 1.5): the hub rows are summed per claim, then across claims -- within 1e-6 (fp32) / 1e-13 (fp64)
    relative of the f64-accumulated product on every row."""
    rng = np.random.default_rng(2)
    d = rng.integers(0, 30, size=6000)
    d[17] = 150_000
    d[2048] = 40_000
    off, idx, val = _csr(d, 1 << 18, seed=9, exact=False)
    val = val.astype(dtype)
    x = (rng.random(1 << 18) + 0.5).astype(dtype)
    got = _run(off, idx, val, d.size, 1 << 18, x)
    prod = val.astype(np.float64) * x.astype(np.float64)[idx]
    want = np.add.reduceat(np.concatenate([prod, [0.0]]), np.minimum(off[:-1], prod.size))
    want[np.diff(off) == 0] = 0.0
    tol = 1e-6 if dtype == np.float32 else 1e-13
    assert np.all(np.abs(got - want) <= tol * np.maximum(np.abs(want), 1e-30)), float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))


def test_rmat_generator_order_small():
    """A Graph500 R-MAT graph with the hub vertices at the low ids (the class the shared-out groups exist for), scale 18."""
    from loops_amd import generate as G
    from oracle import oracle as O
    off, idx, val = G.rmat_csr(18, 16, relabel="none")
    rows = cols = 1 << 18
    x = G.uniform_distribution_int(cols)
    want = O.spmv_f32(off, idx, val, x)
    assert int(np.diff(off.astype(np.int64))[:256].sum()) > 24 * TILE                 # (the first group is heavy)
    for sched in ("group_mapped", "work_oriented"):
        assert np.array_equal(_run(off, idx, val, rows, cols, x, sched), want), sched


def test_memo_survives_a_change_of_content_under_the_same_pointers():
    """The one-shot entry remembers (offsets pointer, rows, nnz) -> "nothing was published" and then runs the one-kernel form.  A
    matrix whose content changes in place (same pointers, same sizes) so that a heavy group appears must still give the right
    product (its owner sweeps it alone); and the other way round."""
    from loops_amd import generate as G, spmv as S
    from oracle import oracle as O
    rows, cols = 3000, 1 << 16
    flat = np.full(rows, 20, np.int64)                                               # 60 000 nonzeros, no heavy group
    hubby = np.full(rows, 1, np.int64); hubby[1000] = 60_000 - (rows - 1)            # same nnz, one hub row of 57 001: 28 tiles
    assert flat.sum() == hubby.sum()
    a, b = _csr(flat, cols, 3), _csr(hubby, cols, 4)
    x = G.uniform_distribution_int(cols)
    csr = S.CSR.from_numpy(rows, cols, *a)
    xd = torch.from_numpy(x).cuda()
    y = torch.empty(rows, device="cuda")
    for content in (a, b, a, b, b, a):
        csr.offsets.copy_(torch.from_numpy(content[0]))
        csr.indices.copy_(torch.from_numpy(content[1]))
        csr.values.copy_(torch.from_numpy(content[2]))
        want = O.spmv_f32(*content, x)
        for _ in range(3):
            y.fill_(-1.0)
            S.spmv("group_mapped", csr, xd, y)
            torch.cuda.synchronize()                                                 # (lets the memo word arrive: the next call reads it)
            assert np.array_equal(y.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("window", [None, 4096])
@pytest.mark.parametrize("hub", [0, 120_000])
def test_gather_order_decided_on_the_device(dtype, window, hub):
    """An x of 6 MB or more and 2^20 nonzeros or more: the launcher samples the columns and runs the builds of the same kernels that
    gather in phases where the sample says "scattered" (uniform columns) and plainly where it does not (a band of 4 096 columns) --
    with a heavy group (publish + claims + fix-up) and without (after the first call the memo sends the owner-only launch).
    Same loads in another order: bit-exact against the oracle either way, call after call."""
    from loops_amd import generate as G, spmv as S
    from oracle import oracle as O
    cols = (1 << 21) + 77 if dtype == np.float32 else (1 << 20) + 77                  # x = 8 MB
    rows = 60_000
    rng = np.random.default_rng(5)
    d = rng.integers(8, 40, size=rows)
    if hub:
        d[257] = hub
    off, idx, val = G.csr_from_degrees(d.astype(np.int64), cols, 11, 0, True, window)
    assert off[-1] >= 1 << 20
    val = val.astype(dtype)
    x = G.uniform_distribution_int(cols).astype(dtype)
    want = O.spmv_f32(off, idx, val.astype(np.float32), x.astype(np.float32)).astype(dtype)   # (exactly summable: the same in both precisions)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    xd = torch.from_numpy(x).cuda()
    y = torch.empty(rows, dtype=xd.dtype, device="cuda")
    for _ in range(4):
        y.fill_(-1.0)
        S.spmv("group_mapped", csr, xd, y)
        torch.cuda.synchronize()                                                     # (the memo word arrives: later calls may take the owner-only launch)
        assert np.array_equal(y.cpu().numpy(), want)
