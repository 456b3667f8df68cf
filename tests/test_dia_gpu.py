"""DIA SpMV through the C ABI: the reference-shaped lane-per-row kernel and the tuned four-rows-per-lane
kernel over the reference's layout (diag_offsets ascending, values column-major [num_diagonals x stride],
container/dia.hxx:69-230), against the CPU oracle on the CSR the diagonals were taken from."""
import numpy as np
import pytest

from conftest import battery, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dia(rows, cols, off, idx, val, stride=None, dtype=np.float32):
    """What dia_t(csr) builds (container/dia.hxx:117-188): distinct (col - row), ascending; zero-filled cells."""
    stride = rows if stride is None else stride
    r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(off))
    d = idx.astype(np.int64) - r
    diags = np.unique(d)
    cells = np.zeros((diags.size, stride), dtype)
    cells[np.searchsorted(diags, d), r] = val
    return diags.astype(np.int32), np.ascontiguousarray(cells)


@pytest.mark.parametrize("tuned", [False, True])
def test_battery(tuned):
    from loops_amd import spmv as S
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        if r == 0:
            continue
        diags, cells = _dia(r, c, off, idx, val)
        x = torch.from_numpy(g[f"{name}.x_int"]).cuda()
        y = torch.full((r,), 7.0, device="cuda")
        S.dia_spmv(r, c, torch.from_numpy(diags).cuda(), torch.from_numpy(cells).cuda(), x, y, tuned=tuned)
        ref, l1 = g[f"{name}.y_int"], g[f"{name}.l1_int"]
        assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= 2e-6 * l1 + 1e-30), (name, tuned)


@pytest.mark.parametrize("rows,cols", [(4096, 4096), (4099, 4099), (1000, 1777), (1777, 1000), (3, 3)])
def test_banded_bit_exact(rows, cols):
    """Band matrices (the format's use case) incl. row counts that are not multiples of 4 (tail lane, unaligned
    strides -> scalar path), rectangular shapes (diagonals leaving the matrix on either side), a padded stride,
    more diagonals than the in-flight batch, fp32 and fp64; exactly-summable inputs -> bit-exact."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rng = np.random.default_rng(rows + cols)
    offs = np.unique(np.concatenate([rng.integers(-min(rows, 40) + 1, min(cols, 40), size=21), [0]]))
    rr, cc = [], []
    for o in offs:
        r = np.arange(max(0, -o), min(rows, cols - o))
        keep = rng.random(r.size) < 0.8
        rr.append(r[keep]); cc.append(r[keep] + o)
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    order = np.lexsort((cc, rr))
    rr, cc = rr[order], cc[order]
    off = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=rows))]).astype(np.int32)
    idx = cc.astype(np.int32)
    val = (rng.integers(1, 9, size=idx.size) / 8.0).astype(np.float32)
    xh = G.uniform_distribution_int(cols)
    want = O.spmv_f32(off, idx, val, xh)
    want64 = O.spmv_f64(off, idx, val.astype(np.float64), xh.astype(np.float64))
    for stride in (rows, (rows + 7) // 4 * 4):
        diags, cells = _dia(rows, cols, off, idx, val, stride)
        for tuned in (False, True):
            y = S.dia_spmv(rows, cols, torch.from_numpy(diags).cuda(), torch.from_numpy(cells).cuda(), torch.from_numpy(xh).cuda(),
                           stride=stride, tuned=tuned).cpu().numpy()
            assert np.array_equal(y, want), (stride, tuned)
            y = S.dia_spmv(rows, cols, torch.from_numpy(diags).cuda(), torch.from_numpy(cells.astype(np.float64)).cuda(),
                           torch.from_numpy(xh.astype(np.float64)).cuda(), stride=stride, tuned=tuned).cpu().numpy()
            assert np.array_equal(y, want64), (stride, tuned, "f64")
