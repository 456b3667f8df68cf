"""Pins the oracle's restatement of the reference's DEVICE-ONLY schedule code against the
reference's own HIP device path executed on the MI355X (oracle/_ref/libloops_ref_gpu.so, built
in the dev container from /root/reference in place; shipped prebuilt to the GPU box)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, battery, load_golden
from loops_amd import _lib

pytestmark = pytest.mark.gpu
SO = os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")


def needs_ref(fn):
    """On a GPU box the reference build MUST be there (oracle/Makefile, dev container): a snapshot without it fails
    instead of skipping, or the device-side pin would silently go untested."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        assert os.path.exists(SO), "oracle/_ref/libloops_ref_gpu.so not shipped: run `make -C oracle ref` in the dev container"
        return fn(*a, **k)
    return wrapper


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _cases():
    from loops_amd import generate as G
    cases = dict(battery())
    rows = cols = 1 << 12
    deg = G.powerlaw_degrees(rows, 1 << 16, cap=1 << 11)
    cases["powerlaw4096"] = (rows, cols) + G.powerlaw_csr(rows, cols, 1 << 16, degrees=deg)
    return cases


@needs_ref
def test_reference_device_merge_path_matches_oracle():
    from oracle import oracle as O
    R = _lib.load_shared(SO)
    for name, (r, c, off, idx, val) in _cases().items():
        nnz = idx.size
        for cfg, (tpb, ipt) in enumerate([(256, 8), (128, 7), (4, 2)]):
            M = O.merge_path_num_tiles(r, nnz, tpb, ipt)
            ts = np.zeros((max(M * tpb, 1), 2), np.uint32)
            owner = np.full(max(nnz, 1), -1, np.int32)
            row = np.full(max(nnz, 1), -1, np.int32)
            vis = np.zeros(max(nnz, 1), np.int32)
            assert R.refgpu_merge_path_dump(cfg, C.c_long(r), C.c_long(nnz), _p(off), _p(ts), _p(owner), _p(row), _p(vis)) == 0
            want = O.merge_path_assign(off, tpb, ipt)
            assert np.array_equal(ts[: M * tpb], want[0]), (name, tpb, ipt)
            assert np.array_equal(owner[:nnz], want[1]) and np.array_equal(row[:nnz], want[2])
            assert np.array_equal(vis[:nnz], want[3])
        M = O.merge_path_num_tiles(r, nnz, 256, 8)
        coords = np.zeros((M + 1, 2), np.uint32)
        assert R.refgpu_merge_path_coords(C.c_long(r), C.c_long(nnz), _p(off), _p(coords)) == 0
        assert np.array_equal(coords, O.merge_path_coords(off, 256, 8)), name


@needs_ref
def test_reference_device_work_oriented_matches_oracle():
    from oracle import oracle as O
    R = _lib.load_shared(SO)
    for name, (r, c, off, idx, val) in _cases().items():
        nnz = idx.size
        for grid in (1, 3, 64):
            tm = np.zeros((grid * 256, 4), np.int32)
            owner = np.full(max(nnz, 1), -1, np.int32)
            row = np.full(max(nnz, 1), -1, np.int32)
            vis = np.zeros(max(nnz, 1), np.int32)
            assert R.refgpu_work_oriented_dump(C.c_long(r), C.c_long(nnz), _p(off), grid, _p(tm), _p(owner), _p(row), _p(vis)) == 0
            want = O.work_oriented_assign(off, grid * 256)
            assert np.array_equal(tm, want[0]), (name, grid)
            assert np.array_equal(owner[:nnz], want[1]) and np.array_equal(row[:nnz], want[2])
            assert np.array_equal(vis[:nnz], want[3])


@needs_ref
def test_reference_device_spmv_matches_golden():
    """The reference's own HIP kernels on this GPU reproduce the golden y (sanity of the pin)."""
    R = _lib.load_shared(SO)
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        if r == 0 or idx.size == 0:
            continue
        x = g[name + ".x_int"]
        for kind in (0, 1, 2):
            y = np.zeros(r, np.float32)
            ms = C.c_float()
            assert R.refgpu_spmv_f32(kind, C.c_long(r), C.c_long(c), C.c_long(idx.size), _p(off), _p(idx), _p(val),
                                     _p(x), _p(y), 1, C.byref(ms)) == 0
            assert np.allclose(y, g[name + ".y_int"], rtol=1e-4, atol=1e-3), (name, kind)


@needs_ref
def test_other_formats_match_the_reference_kernels():
    """COO / CSC / ELL (thread-mapped and ell_merge_path) / DIA: our tuned kernels against the REFERENCE'S kernels for those formats, executed on this
    GPU from containers built by the reference's own converting constructors; BCSR 4x4 against its
    bcsr_thread_mapped<4, 4> on the same block arrays.  Exactly-summable inputs: bit-exact."""
    import ctypes as C
    import torch
    from loops_amd import _lib, generate as G, spmv as S
    R = _lib.load_shared(SO)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rows, cols = 3000, 5000
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 40, size=rows)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) for n in lens]).astype(np.int32)
    val = (rng.integers(1, 9, size=idx.size) / 8.0).astype(np.float32)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(off))
    ours = {
        0: S.coo_spmv(rows, cols, torch.from_numpy(ri).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(val).cuda(), x),
    }
    order = np.lexsort((ri, idx))
    coff = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=cols))]).astype(np.int32)
    ours[1] = S.csc_spmv(rows, cols, torch.from_numpy(coff).cuda(), torch.from_numpy(ri[order]).cuda(),
                         torch.from_numpy(val[order]).cuda(), x)
    pitch = int(lens.max())
    ind = np.full((rows, pitch), -1, np.int32)
    ev = np.zeros((rows, pitch), np.float32)
    for r in range(rows):
        ind[r, :lens[r]] = idx[off[r]:off[r + 1]]
        ev[r, :lens[r]] = val[off[r]:off[r + 1]]
    ours[2] = S.ell_spmv(rows, cols, pitch, torch.from_numpy(ind).cuda(), torch.from_numpy(ev).cuda(), x)
    ours[4] = S.ell_spmv(rows, cols, pitch, torch.from_numpy(ind).cuda(), torch.from_numpy(ev).cuda(), x, tuned="merge_path")
    for fmt in (0, 1, 2, 4):
        yr = np.zeros(rows, np.float32)
        ms = C.c_float()
        rc = R.refgpu_format_spmv_f32(fmt, C.c_long(rows), C.c_long(cols), C.c_long(idx.size), p(off), p(idx), p(val), p(xh), p(yr),
                                      1, C.byref(ms))
        assert rc == 0 and np.array_equal(ours[fmt].cpu().numpy(), yr), fmt
    # DIA: the reference's dia_t(csr) + dia_thread_mapped on a banded matrix vs our container-free kernels on the same
    # diagonals (built here as dia.hxx:117-188 builds them)
    rng = np.random.default_rng(10)
    n = 4099
    offs = np.unique(np.concatenate([rng.integers(-30, 31, size=15), [0]]))
    rr, cc = [], []
    for o in offs:
        r = np.arange(max(0, -o), min(n, n - o))
        keep = rng.random(r.size) < 0.7
        rr.append(r[keep]); cc.append(r[keep] + o)
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    order = np.lexsort((cc, rr))
    doff = np.concatenate([[0], np.cumsum(np.bincount(rr[order], minlength=n))]).astype(np.int32)
    didx = cc[order].astype(np.int32)
    dval = (rng.integers(1, 9, size=didx.size) / 8.0).astype(np.float32)
    dx = G.uniform_distribution_int(n)
    d = didx.astype(np.int64) - np.repeat(np.arange(n, dtype=np.int64), np.diff(doff))
    diags = np.unique(d)
    cells = np.zeros((diags.size, n), np.float32)
    cells[np.searchsorted(diags, d), np.repeat(np.arange(n), np.diff(doff))] = dval
    yr = np.zeros(n, np.float32)
    ms = C.c_float()
    rc = R.refgpu_format_spmv_f32(3, C.c_long(n), C.c_long(n), C.c_long(didx.size), p(doff), p(didx), p(dval), p(dx), p(yr), 1,
                                  C.byref(ms))
    assert rc == 0
    for tuned in (False, True):
        yo = S.dia_spmv(n, n, torch.from_numpy(diags.astype(np.int32)).cuda(), torch.from_numpy(cells).cuda(),
                        torch.from_numpy(dx).cuda(), tuned=tuned).cpu().numpy()
        assert np.array_equal(yo, yr), ("dia", tuned)
    nbr = 1 << 10
    boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, 16)
    xb = G.uniform_distribution_int(nbr * 4)
    b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    yb = S.bcsr_thread_mapped(b, torch.from_numpy(xb).cuda(), mfma=1).cpu().numpy()
    yr = np.zeros(nbr * 4, np.float32)
    ms = C.c_float()
    rc = R.refgpu_bcsr4x4_spmv_f32(C.c_long(nbr * 4), C.c_long(nbr * 4), C.c_long(nbr), C.c_long(nbr), C.c_long(bcols.size),
                                   p(boff), p(bcols), p(bvals), p(xb), p(yr), 1, C.byref(ms))
    assert rc == 0 and np.array_equal(yb, yr)
