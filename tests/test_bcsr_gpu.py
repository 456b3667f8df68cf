"""BCSR SpMV for the block shapes the reference ships and tests (examples/spmv/bcsr_thread_mapped.cu:31-32: 2 x 2;
unittests/test_spmv_bcsr.cu:24-36: 2 x 2 and 3 x 3; every example also as .f64) plus 4 x 4 and 8 x 8, fp32 and fp64:
the coalesced lane-group kernels (loops_spmv_bcsr_* mode 2 / 3, kernels/bcsr_spmv.hxx) against the oracle, against a
numpy block product, and against the REFERENCE'S OWN bcsr_thread_mapped<R, R> executed on this GPU (oracle/_ref).
Inputs are exactly summable (cells k/8, integer x), so every comparison is BIT-EXACT whatever the summation order."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")
SHAPES = [2, 3, 4, 8]
EXPLICIT = tuple(100000 + 100 * h + u for h in (1, 4, 16) for u in (1, 2, 4))  # every compiled kernel shape


def _blocks(R, nbr, nbc, lens, seed, dtype):
    rng = np.random.default_rng(seed)
    boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    bcols = np.concatenate([np.sort(rng.choice(nbc, size=int(n), replace=False)) for n in lens]
                           + [np.zeros(0, np.int64)]).astype(np.int32)
    bvals = (rng.integers(-8, 9, size=bcols.size * R * R) / 8.0).astype(dtype)
    x = rng.integers(1, 11, size=nbc * R).astype(dtype)
    return boff, bcols, bvals, x


def _numpy_product(R, rows, boff, bcols, bvals, x):
    nbr = boff.size - 1
    y = np.zeros(nbr * R, np.float64)
    blocks = bvals.reshape(-1, R, R).astype(np.float64)
    xs = x.reshape(-1, R).astype(np.float64)
    prod = np.einsum("bij,bj->bi", blocks, xs[bcols])            # exact: cells k/8, x small integers
    np.add.at(y.reshape(nbr, R), np.repeat(np.arange(nbr), np.diff(boff)), prod)
    return y[:rows].astype(bvals.dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("R", SHAPES)
def test_every_mode_matches_the_block_product(R, dtype):
    from loops_amd import spmv as S
    from oracle import oracle as O
    cases = {
        "uniform16": (700, 900, np.full(700, 16)),
        "ragged": (1000, 1000, np.random.default_rng(R).integers(0, 41, size=1000)),          # 0 .. 40 blocks: no multiple of any h
        "short": (3000, 500, np.random.default_rng(R + 1).integers(0, 3, size=3000)),          # mean ~1: h = 1
        "one_long_row": (65, 4096, np.concatenate([[3000], np.random.default_rng(R + 2).integers(0, 5, size=64)])),
    }
    for name, (nbr, nbc, lens) in cases.items():
        boff, bcols, bvals, x = _blocks(R, nbr, nbc, lens, seed=10 * R + len(name), dtype=dtype)
        rows = nbr * R - (R - 1)                                   # last block-row partly outside the matrix: guarded stores
        want = _numpy_product(R, rows, boff, bcols, bvals, x)
        if dtype == np.float32:
            assert np.array_equal(want, O.bcsr_spmv_f32(R, R, rows, boff, bcols, bvals, x)), (name, "oracle")
        b = S.BCSR(R, R, rows, nbc * R, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
        xd = torch.from_numpy(x).cuda()
        modes = ("thread", "coalesced", "tuned") + EXPLICIT + (("mfma", "merge_path") if R == 4 and dtype == np.float32 else ())
        for mode in modes:
            y = torch.full((rows + R,), 7.0, dtype=xd.dtype, device="cuda")   # rows >= `rows` must stay untouched
            S.bcsr_thread_mapped(b, xd, y, mfma=mode)
            got = y.cpu().numpy()
            assert np.array_equal(got[:rows], want), (R, dtype.__name__, name, mode)
            assert np.all(got[rows:] == 7.0), (R, dtype.__name__, name, mode, "wrote past `rows`")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("R", SHAPES)
def test_matrix_without_blocks(R, dtype):
    """num_block_rows > 0, num_blocks == 0 (null block arrays): y = 0 from every mode, nothing is dereferenced."""
    from loops_amd import spmv as S
    nbr = 37
    boff = torch.zeros(nbr + 1, dtype=torch.int32, device="cuda")
    empty_i = torch.zeros(0, dtype=torch.int32, device="cuda")
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    b = S.BCSR(R, R, nbr * R - 1, nbr * R, boff, empty_i, torch.zeros(0, dtype=tdt, device="cuda"))
    x = torch.ones(nbr * R, dtype=tdt, device="cuda")
    for mode in ("thread", "coalesced", "tuned") + (("mfma", 144) if R == 4 and dtype == np.float32 else ()):
        y = torch.full((nbr * R - 1,), 5.0, dtype=tdt, device="cuda")
        S.bcsr_thread_mapped(b, x, y, mfma=mode)
        assert torch.count_nonzero(y).item() == 0, (R, mode)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("R", SHAPES)
def test_against_the_reference_kernel_on_this_gpu(R, dtype):
    """The reference's own bcsr_thread_mapped<R, R> (compiled in place from /root/reference, oracle/ref_gpu_shim.cpp)
    on the same block arrays, same GPU."""
    assert os.path.exists(REF_SO), "oracle/_ref/libloops_ref_gpu.so missing: run __graft_entry__.build() where /root/reference exists"
    from loops_amd import _lib, spmv as S
    ref = _lib.load_shared(REF_SO)
    assert hasattr(ref, "refgpu_bcsr_spmv"), "oracle/_ref/libloops_ref_gpu.so is stale (no refgpu_bcsr_spmv): rebuild it"
    nbr = nbc = 2048
    lens = np.random.default_rng(100 + R).integers(0, 33, size=nbr)
    boff, bcols, bvals, x = _blocks(R, nbr, nbc, lens, seed=R, dtype=dtype)
    rows = nbr * R
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    yr = np.zeros(rows, dtype)
    ms = C.c_float()
    rc = ref.refgpu_bcsr_spmv(R, int(dtype == np.float64), C.c_long(rows), C.c_long(nbc * R), C.c_long(nbr), C.c_long(nbc),
                              C.c_long(bcols.size), p(boff), p(bcols), p(bvals), p(x), p(yr), 1, C.byref(ms))
    assert rc == 0
    assert np.array_equal(yr, _numpy_product(R, rows, boff, bcols, bvals, x))
    b = S.BCSR(R, R, rows, nbc * R, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    for mode in ("coalesced", "tuned"):
        got = S.bcsr_thread_mapped(b, torch.from_numpy(x).cuda(), mfma=mode).cpu().numpy()
        assert np.array_equal(got, yr), (R, dtype.__name__, mode)


def test_real_valued_blocks_within_the_fp32_bound():
    """Real-valued cells and x: |y - y64| <= 2e-6 * sum |a x| per row (the reference's Wilkinson-style criterion,
    util/reference.hxx:278-337, at the tolerance measured for the CSR kernels)."""
    from loops_amd import spmv as S
    rng = np.random.default_rng(5)
    for R in SHAPES:
        nbr = nbc = 1500
        lens = rng.integers(0, 25, size=nbr)
        boff, bcols, _, _ = _blocks(R, nbr, nbc, lens, seed=R, dtype=np.float32)
        bvals = rng.uniform(-1.0, 1.0, size=bcols.size * R * R).astype(np.float32)
        x = rng.uniform(-1.0, 1.0, size=nbc * R).astype(np.float32)
        blocks, xs = bvals.reshape(-1, R, R).astype(np.float64), x.reshape(-1, R).astype(np.float64)
        y64, l1 = np.zeros((nbr, R)), np.zeros((nbr, R))
        owner = np.repeat(np.arange(nbr), np.diff(boff))
        np.add.at(y64, owner, np.einsum("bij,bj->bi", blocks, xs[bcols]))
        np.add.at(l1, owner, np.einsum("bij,bj->bi", np.abs(blocks), np.abs(xs[bcols])))
        b = S.BCSR(R, R, nbr * R, nbc * R, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
        for mode in ("thread", "tuned"):
            got = S.bcsr_thread_mapped(b, torch.from_numpy(x).cuda(), mfma=mode).cpu().numpy().astype(np.float64)
            assert np.all(np.abs(got - y64.ravel()) <= 2e-6 * l1.ravel() + 1e-30), (R, mode)


def test_merge_path_bcsr_on_skewed_block_rows():
    """The merge-path form of the 4 x 4 fp32 product (loops_spmv_bcsr_f32 mode 4, kernels/bcsr_merge_path.hxx): block-rows of several
    tiles (carry-outs across many tiles), hub block-rows next to empty ones, tiles that hold a thousand short block-rows, a matrix
    that ends in a hub; bit-exact against the block product; rows past `rows` untouched; repeated calls share the per-stream scratch."""
    from loops_amd import spmv as S
    rng = np.random.default_rng(12)
    cases = {
        "hubs": np.concatenate([[5000], rng.integers(0, 4, size=300), [0] * 40, [9000, 2500], rng.integers(0, 9, size=2000), [7000]]),
        "short": rng.integers(0, 2, size=6000),
        "empty_front": np.concatenate([np.zeros(1500, np.int64), [3000], np.zeros(1500, np.int64)]),
        "one_tile": np.array([3, 0, 5]),
    }
    for name, lens in cases.items():
        nbr, nbc = lens.size, 12000
        boff, bcols, bvals, x = _blocks(4, nbr, nbc, lens, seed=len(name), dtype=np.float32)
        rows = nbr * 4 - 2
        want = _numpy_product(4, rows, boff, bcols, bvals, x)
        b = S.BCSR(4, 4, rows, nbc * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
        xd = torch.from_numpy(x).cuda()
        for _ in range(3):
            y = torch.full((rows + 4,), 7.0, device="cuda")
            S.bcsr_thread_mapped(b, xd, y, mfma="merge_path")
            got = y.cpu().numpy()
            assert np.array_equal(got[:rows], want), name
            assert np.all(got[rows:] == 7.0), name


def _class_by_the_rule(lens):
    """numpy restatement of kernels::bcsr_row_length_class (bcsr_merge_path.hxx): skewed when the longest block-row exceeds
    max(64, blocks / 14 000) or groups of 4 consecutive block-rows walked in lockstep touch more than 1.2 x the blocks."""
    lens = np.asarray(lens, np.int64)
    nb = int(lens.sum())
    if lens.size == 0 or nb == 0:
        return "even"
    pad = np.concatenate([lens, np.zeros((-lens.size) % 4, np.int64)]).reshape(-1, 4)
    lockstep = int(4 * pad.max(axis=1).sum())
    return "skewed" if (int(lens.max()) > max(64, nb // 14000) or 10 * lockstep > 12 * nb) else "even"


def test_tuned_mode_picks_the_kernel_by_the_block_row_lengths():
    """loops_spmv_bcsr_f32 mode "tuned" on 4 x 4 fp32: the first call on a matrix runs the merge-path tiles and launches the probe of
    the block-row lengths; once its report has arrived later calls run the MFMA kernel on even lengths and stay on the tiles on
    skewed ones (include/loops_amd.h).  Whatever ran: bit-exact against the block product, call after call, matrices alternating;
    loops_bcsr_row_length_class agrees with the numpy restatement of the rule."""
    from loops_amd import spmv as S
    rng = np.random.default_rng(21)
    cases = {
        "even16": np.full(5000, 16),
        "mild": rng.integers(14, 19, size=5000),                 # lockstep ~1.1 x
        "spread": rng.integers(8, 25, size=5000),                # lockstep ~1.3 x
        "hub": np.concatenate([rng.integers(0, 9, size=3000), [6000], rng.integers(0, 9, size=3000)]),
        "alternating": np.tile([0, 0, 0, 40], 1500),            # lockstep groups of 4 touch 4 x the blocks
        "row_of_65": np.concatenate([np.full(4000, 12), [65]]),
        "row_of_64": np.concatenate([np.full(4000, 12), [64]]),
    }
    built = {}
    for name, lens in cases.items():
        nbr, nbc = lens.size, 8000
        boff, bcols, bvals, x = _blocks(4, nbr, nbc, lens, seed=len(name) + 3, dtype=np.float32)
        rows = nbr * 4 - 1
        b = S.BCSR(4, 4, rows, nbc * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
        built[name] = (b, torch.from_numpy(x).cuda(), _numpy_product(4, rows, boff, bcols, bvals, x), rows)
        assert S.bcsr_row_length_class(b) == _class_by_the_rule(lens), name
    assert [_class_by_the_rule(cases[n]) for n in ("even16", "mild", "spread", "hub", "alternating", "row_of_65", "row_of_64")] == \
        ["even", "even", "skewed", "skewed", "skewed", "skewed", "even"]
    for _ in range(3):
        for name, (b, xd, want, rows) in built.items():
            for _ in range(2):
                y = torch.full((rows + 4,), 7.0, device="cuda")
                S.bcsr_thread_mapped(b, xd, y, mfma="tuned")
                torch.cuda.synchronize()                                       # (the probe's report arrives: the next call may change kernel)
                got = y.cpu().numpy()
                assert np.array_equal(got[:rows], want), name
                assert np.all(got[rows:] == 7.0), name


def test_tuned_mode_from_two_streams_and_a_captured_graph():
    """Mode "tuned" reads its memo on the host when the call is made and launches on the caller's stream only: two streams at once, and
    a HIP graph captured on a stream the call has run on before (scratch and memo exist then), must give the block product."""
    from loops_amd import spmv as S
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(0, 9, size=4000), [5000], rng.integers(0, 9, size=4000)])
    nbr, nbc = lens.size, 9000
    boff, bcols, bvals, x = _blocks(4, nbr, nbc, lens, seed=2, dtype=np.float32)
    rows = nbr * 4
    want = torch.from_numpy(_numpy_product(4, rows, boff, bcols, bvals, x)).cuda()
    b = S.BCSR(4, 4, rows, nbc * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    xd = torch.from_numpy(x).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ys = [torch.full((rows,), -1.0, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    for _ in range(4):
        for st, y in zip(streams, ys):
            with torch.cuda.stream(st):
                S.bcsr_thread_mapped(b, xd, y, mfma="tuned")
    torch.cuda.synchronize()
    assert torch.equal(ys[0], want) and torch.equal(ys[1], want)
    side = streams[0]
    yg = torch.empty(rows, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        S.bcsr_thread_mapped(b, xd, yg, mfma="tuned")
    for _ in range(3):
        yg.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg, want)
