"""ELL SpMV through the C ABI: the reference-shaped lane-per-row kernel and the tuned
row-split kernel over the reference's row-major rows x pitch layout (padding = column -1,
container/ell.hxx:31-55), against the CPU oracle on the same matrix."""
import numpy as np
import pytest

from conftest import battery, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _ell(off, idx, val, pitch=None):
    rows = off.size - 1
    lens = np.diff(off)
    pitch = int(lens.max()) if pitch is None and rows else (pitch or 0)
    ind = np.full((rows, max(pitch, 1)), -1, np.int32)[:, :pitch]
    v = np.zeros((rows, pitch), np.float32)
    for r in range(rows):
        n = lens[r]
        ind[r, :n] = idx[off[r]:off[r + 1]]
        v[r, :n] = val[off[r]:off[r + 1]]
    return pitch, np.ascontiguousarray(ind), np.ascontiguousarray(v)


@pytest.mark.parametrize("tuned", [False, True, "merge_path"])
def test_battery(tuned):
    from loops_amd import spmv as S
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        pitch, ind, v = _ell(off, idx, val)
        x = torch.from_numpy(g[f"{name}.x_int"]).cuda()
        y = torch.full((r,), 7.0, device="cuda")
        S.ell_spmv(r, c, pitch, torch.from_numpy(ind).cuda(), torch.from_numpy(v).cuda(), x, y, tuned=tuned)
        ref, l1 = g[f"{name}.y_int"], g[f"{name}.l1_int"]
        assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= 2e-6 * l1 + 1e-30), (name, tuned)


@pytest.mark.parametrize("pitch_pad", [0, 1, 3, 5])
def test_regular_matrix_bit_exact_every_group_width(pitch_pad):
    """Rows of 1..N nonzeros padded to pitches that are / are not multiples of 4 (vector and scalar
    paths), every sub-group width G = 1..64; exactly-summable inputs -> bit-exact."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rng = np.random.default_rng(pitch_pad)
    rows, cols = 3000, 5000
    xh = G.uniform_distribution_int(cols)
    for maxlen in (1, 3, 8, 16, 33, 70, 150, 300):
        lens = rng.integers(0, maxlen + 1, size=rows)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        idx = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        val = (rng.integers(1, 9, size=idx.size) / 8.0).astype(np.float32)
        pitch, ind, v = _ell(off, idx, val, int(lens.max()) + pitch_pad)
        want = O.spmv_f32(off, idx, val, xh)
        for tuned in (False, True, "merge_path"):
            y = S.ell_spmv(rows, cols, pitch, torch.from_numpy(ind).cuda(), torch.from_numpy(v).cuda(),
                           torch.from_numpy(xh).cuda(), tuned=tuned).cpu().numpy()
            assert np.array_equal(y, want), (maxlen, pitch, tuned)
        # fp64 twins of the three paths
        want64 = O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64))
        for tuned in (False, True, "merge_path"):
            y = S.ell_spmv(rows, cols, pitch, torch.from_numpy(ind).cuda(), torch.from_numpy(v.astype(np.float64)).cuda(),
                           torch.from_numpy(xh.astype(np.float64)).cuda(), tuned=tuned).cpu().numpy()
            assert np.array_equal(y, want64), (maxlen, pitch, tuned, "f64")


def test_ell_merge_path_engine_padding_and_long_rows():
    """algorithms::spmv::ell_merge_path on the fused engine: rows longer than a merge tile (carry-outs + fix-up),
    pitches that leave the cell arrays unaligned for 16-byte loads, padding cells whose x would be poison (NaN at
    x[0] and at the largest index: a padding cell must never multiply), y not pre-zeroed, empty rows."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    for rows, cols, maxlen, pad in ((700, 9000, 6000, 0), (4000, 3000, 9, 2), (257, 20000, 9000, 3), (5, 5, 0, 2)):
        lens = rng.integers(0, maxlen + 1, size=rows)
        lens[::7] = 0
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        idx = np.concatenate([np.sort(rng.choice(np.arange(1, cols), size=n, replace=False)) for n in lens]
                             + [np.zeros(0, np.int64)]).astype(np.int32)   # column 0 is never a real column here
        val = (rng.integers(1, 9, size=idx.size) / 8.0).astype(np.float32)
        xh = G.uniform_distribution_int(cols)
        xh[0] = np.nan  # what a padding cell's clamped gather would read
        pitch, ind, v = _ell(off, idx, val, int(lens.max(initial=0)) + pad)
        v[ind < 0] = np.inf  # padding VALUES are garbage too: only the column sign may decide
        want = O.spmv_f32(off, idx, val, xh)
        y = torch.full((rows,), np.nan, device="cuda")
        S.ell_spmv(rows, cols, pitch, torch.from_numpy(ind).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(xh).cuda(), y,
                   tuned="merge_path")
        assert np.array_equal(y.cpu().numpy(), want), (rows, cols, maxlen, pad)

