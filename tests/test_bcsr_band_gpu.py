"""Block-band plan for 4 x 4 fp32 BCSR (loops_bcsr_band_plan_*, include/loops/kernels/bcsr_band.hxx): the product of
algorithms::spmv::bcsr_thread_mapped<4, 4> (reference bcsr_thread_mapped.cuh:36-74) over a re-ordered copy of the blocks.
Inputs are exactly summable (cells k/8, integer x), so every comparison with the oracle's BCSR restatement, with a numpy
block product and with the shipped BCSR kernels is BIT-EXACT whatever the summation order; the layout arrays are compared
with a numpy restatement of the documented order (band, block column, BCSR position)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _blocks(nbr, nbc, lens, seed, cells="eighths"):
    rng = np.random.default_rng(seed)
    boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    bcols = np.concatenate([np.sort(rng.choice(nbc, size=int(n), replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    bvals = (rng.integers(-8, 9, size=bcols.size * 16) / 8.0).astype(np.float32)
    x = rng.integers(1, 11, size=nbc * 4).astype(np.float32)
    return boff, bcols, bvals, x


def _numpy_product(rows, boff, bcols, bvals, x):
    nbr = boff.size - 1
    y = np.zeros(nbr * 4, np.float64)
    prod = np.einsum("bij,bj->bi", bvals.reshape(-1, 4, 4).astype(np.float64), x.reshape(-1, 4).astype(np.float64)[bcols])
    np.add.at(y.reshape(nbr, 4), np.repeat(np.arange(nbr), np.diff(boff)), prod)
    return y[:rows].astype(np.float32)


CASES = {
    "uniform16": (700, 900, lambda r: np.full(700, 16)),
    "ragged": (1000, 1000, lambda r: r.integers(0, 41, size=1000)),
    "short": (3000, 500, lambda r: r.integers(0, 3, size=3000)),
    "one_long_row": (65, 4096, lambda r: np.concatenate([[3000], r.integers(0, 5, size=64)])),
    "empty_bands": (400, 300, lambda r: np.concatenate([np.zeros(130, np.int64), r.integers(1, 9, size=70), np.zeros(200, np.int64)])),
    "one_block_row": (1, 64, lambda r: np.array([40])),
    "hub_block_rows": (600, 2000, lambda r: np.concatenate([[1900, 3, 1500], r.integers(0, 6, size=296), [1200], r.integers(0, 6, size=300)])),
}


def _device(boff, bcols, bvals, rows, nbc):
    from loops_amd import spmv as S
    return S.BCSR(4, 4, rows, nbc * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())


@pytest.mark.parametrize("name", sorted(CASES))
def test_product_matches_oracle_and_block_product(name):
    from loops_amd import spmv as S
    from oracle import oracle as O
    nbr, nbc, lens = CASES[name]
    lens = lens(np.random.default_rng(len(name)))
    boff, bcols, bvals, x = _blocks(nbr, nbc, lens, seed=7 + len(name))
    rows = nbr * 4 - 3                                              # last block-row partly outside the matrix: guarded stores
    want = _numpy_product(rows, boff, bcols, bvals, x)
    assert np.array_equal(want, O.bcsr_spmv_f32(4, 4, rows, boff, bcols, bvals, x)), "oracle"
    b = _device(boff, bcols, bvals, rows, nbc)
    xd = torch.from_numpy(x).cuda()
    for hb, chunks in ((0, 0), (16, 0), (64, 0), (16, 1), (64, 1000), (256, 7), (4096, 0), (4096, 300)):
        plan = S.BCSRBandPlan(b, band_block_rows=hb, target_chunks=chunks)
        assert plan.num_bands == -(-nbr // plan.HB)
        for shape in ((8, 1, 0), (8, 2, 1), (16, 4, 0), (16, 2, 1), (8, 4, 1)):
            plan.set_shape(*shape)
            y = torch.full((rows + 4,), 7.0, device="cuda")
            plan.spmv(xd, y)
            got = y.cpu().numpy()
            assert np.array_equal(got[:rows], want), (name, hb, chunks, shape)
            assert np.all(got[rows:] == 7.0), (name, hb, chunks, shape, "wrote past `rows`")
        plan.close()


def _hubs_of(lens, hb):
    """numpy restatement of the hub rule: per band the first 32 block-rows (in order) holding >= max(64, band blocks // 32) blocks."""
    nbr = lens.size
    out = {}
    for band in range(-(-nbr // hb)):
        mine = lens[band * hb:(band + 1) * hb]
        threshold = max(64, int(mine.sum()) // 32)
        out[band] = [int(r) for r in np.nonzero(mine >= threshold)[0][:32]]
    return out


def test_layout_arrays_follow_the_documented_order():
    """values / words / perm / hubs against a numpy restatement: blocks sorted by (band, block column, BCSR position), every band
    padded to whole steps of 16 blocks with zero cells, row code HB, column 0, perm -1; hub block-rows coded as replicas picked by the
    slot's place in its step; the work list covers every step once."""
    from loops_amd import spmv as S
    nbr, nbc = 500, 700
    lens = np.random.default_rng(5).integers(0, 30, size=nbr)
    lens[[3, 130, 131, 499]] = [400, 300, 90, 250]                                # hub block-rows (and one just below a band's threshold)
    boff, bcols, bvals, _ = _blocks(nbr, nbc, lens, seed=11)
    b = _device(boff, bcols, bvals, nbr * 4, nbc)
    for hb, chunks in ((64, 0), (16, 40), (128, 300)):
        plan = S.BCSRBandPlan(b, band_block_rows=hb, target_chunks=chunks)
        val, words, perm, chunk_list, multi, hubs = plan.arrays()
        want_hubs = _hubs_of(lens, hb)
        assert any(want_hubs.values())
        br_of = np.repeat(np.arange(nbr), np.diff(boff))
        order = np.lexsort((np.arange(bcols.size), bcols, br_of // hb))
        cmask = (1 << plan.cbits) - 1
        at = 0
        for band in range(plan.num_bands):
            assert hubs[band, 0] == len(want_hubs[band]) and list(hubs[band, 1:1 + hubs[band, 0]]) == want_hubs[band], band
            mine = order[br_of[order] // hb == band]
            n = mine.size
            assert np.array_equal(perm[at:at + n], mine)
            assert np.array_equal(words[at:at + n] & cmask, bcols[mine])
            rows_in = br_of[mine] % hb
            code = rows_in.copy()
            for k, hub_row in enumerate(want_hubs[band]):
                sel = rows_in == hub_row
                code[sel] = hb + 1 + 16 * k + (at + np.nonzero(sel)[0]) % 16
            assert np.array_equal(words[at:at + n] >> plan.cbits, code)
            assert np.array_equal(val[at:at + n].reshape(n, 16), bvals.reshape(-1, 16)[mine])
            padded = -(-n // 16) * 16
            assert np.all(perm[at + n:at + padded] == -1) and np.all(words[at + n:at + padded] == (hb << plan.cbits))
            assert np.all(val[at + n:at + padded] == 0)
            at += padded
        assert at == plan.slots
        covered = np.zeros(plan.steps, np.int32)
        for band, s0, s1, slot in chunk_list:
            covered[s0:s1] += 1
        assert np.all(covered == 1)
        assert sorted(set(chunk_list[:, 0])) == list(range(plan.num_bands))       # (bands without blocks own an empty chunk)
        cut = {int(m[0]): int(m[2]) for m in multi}
        for band in range(plan.num_bands):
            pieces = int(np.sum(chunk_list[:, 0] == band))
            assert pieces == cut.get(band, 1)
        plan.close()


def test_refresh_values_and_tune():
    from loops_amd import spmv as S
    nbr, nbc = 2048, 2048
    boff, bcols, bvals, x = _blocks(nbr, nbc, np.full(nbr, 12), seed=3)
    b = _device(boff, bcols, bvals, nbr * 4, nbc)
    xd = torch.from_numpy(x).cuda()
    plan = S.BCSRBandPlan(b)
    times = plan.tune(3)
    assert len(times) == 12 and all(t > 0 for t in times.values())
    fastest = min(times, key=times.get)                               # (the shape the plan came with stays unless another is 2 % faster)
    assert times[(plan.waves, plan.unroll, plan.nt)] <= 1.03 * times[fastest]
    assert np.array_equal(plan.spmv(xd).cpu().numpy(), _numpy_product(nbr * 4, boff, bcols, bvals, x))
    new_vals = (np.random.default_rng(9).integers(-8, 9, size=bvals.size) / 8.0).astype(np.float32)
    plan.refresh_values(torch.from_numpy(new_vals).cuda())
    assert np.array_equal(plan.spmv(xd).cpu().numpy(), _numpy_product(nbr * 4, boff, bcols, new_vals, x))


def test_degenerate_inputs_and_errors():
    from loops_amd import spmv as S, _lib
    # no block-rows at all / block-rows without any block (null block arrays): nothing is dereferenced, y = 0
    empty_i = torch.zeros(0, dtype=torch.int32, device="cuda")
    empty_f = torch.zeros(0, dtype=torch.float32, device="cuda")
    plan = S.BCSRBandPlan(S.BCSR(4, 4, 0, 0, torch.zeros(1, dtype=torch.int32, device="cuda"), empty_i, empty_f))
    plan.spmv(torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda"))
    nbr = 37
    b = S.BCSR(4, 4, nbr * 4 - 1, nbr * 4, torch.zeros(nbr + 1, dtype=torch.int32, device="cuda"), empty_i, empty_f)
    plan = S.BCSRBandPlan(b)
    y = torch.full((nbr * 4,), 5.0, device="cuda")
    plan.spmv(torch.ones(nbr * 4, device="cuda"), y)
    got = y.cpu().numpy()
    assert np.all(got[:nbr * 4 - 1] == 0) and got[-1] == 5.0
    # a block column outside the matrix, a band height that is not a power of two
    boff, bcols, bvals, _ = _blocks(10, 10, np.full(10, 3), seed=1)
    bad = bcols.copy()
    bad[7] = 10
    with pytest.raises(_lib.LoopsError, match="BADARG"):
        S.BCSRBandPlan(_device(boff, bad, bvals, 40, 10))
    bad[7] = -1
    with pytest.raises(_lib.LoopsError, match="BADARG"):
        S.BCSRBandPlan(_device(boff, bad, bvals, 40, 10))
    with pytest.raises(_lib.LoopsError, match="BADARG"):
        S.BCSRBandPlan(_device(boff, bcols, bvals, 40, 10), band_block_rows=48)
    with pytest.raises(_lib.LoopsError, match="BADARG"):
        S.BCSRBandPlan(_device(boff, bcols, bvals, 40, 10), band_block_rows=8192)


def test_c4_full_size_block_band_bit_exact():
    """BASELINE config C4 at FULL size (2^18 block-rows x 16 blocks, 295 MB): automatic plan (one uncut band of 1024 block-rows per
    compute unit), 4096-block-row bands cut into 256 chunks (partial vectors + combine) and other cuts, bit-exact against the oracle."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    nbr, per = 1 << 18, 16
    boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
    xh = G.uniform_distribution_int(nbr * 4)
    want = O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, xh)
    b = _device(boff, bcols, bvals, nbr * 4, nbr)
    x = torch.from_numpy(xh).cuda()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for hb, chunks in ((0, 0), (4096, 0), (4096, 64), (2048, 512)):
        plan = S.BCSRBandPlan(b, band_block_rows=hb, target_chunks=chunks)
        if hb == 0 and cus == 256:
            assert plan.HB == 1024 and plan.num_bands == 256 and plan.num_multi == 0
        if (hb, chunks) == (4096, 0) and cus == 256:
            assert plan.num_bands == 64 and plan.num_multi == 64 and plan.num_chunks == 256
        y = torch.full((nbr * 4,), -1.0, device="cuda")
        plan.spmv(x, y)
        assert np.array_equal(y.cpu().numpy(), want), (hb, chunks)
        plan.close()
