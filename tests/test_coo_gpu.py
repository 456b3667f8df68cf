"""COO SpMV through the C ABI: the reference-shaped kernel (one atomic per nonzero) and the
tuned run kernel, against the CPU oracle's reference::spmv restatement on the same matrix."""
import numpy as np
import pytest

from conftest import battery, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _coo(off, idx, val):
    rows = np.repeat(np.arange(off.size - 1, dtype=np.int32), np.diff(off))
    return rows, np.asarray(idx, np.int32), np.asarray(val, np.float32)


@pytest.mark.parametrize("tuned", [False, True])
@pytest.mark.parametrize("order", ["sorted", "shuffled"])
def test_battery(tuned, order):
    from loops_amd import spmv as S
    g = load_golden("battery.npz")
    rng = np.random.default_rng(5)
    for name, (r, c, off, idx, val) in battery().items():
        ri, ci, v = _coo(off, idx, val)
        if order == "shuffled":
            p = rng.permutation(ri.size)
            ri, ci, v = ri[p], ci[p], v[p]
        x = torch.from_numpy(g[f"{name}.x_int"]).cuda()
        y = torch.full((r,), 7.0, device="cuda")
        S.coo_spmv(r, c, torch.from_numpy(ri).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda(), x, y, tuned=tuned)
        ref, l1 = g[f"{name}.y_int"], g[f"{name}.l1_int"]
        assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= 2e-6 * l1 + 1e-30), (name, tuned, order)


@pytest.mark.parametrize("shift", [0, 1, 3])
def test_powerlaw_bit_exact_and_unaligned(shift):
    """Exactly-summable inputs: any summation order is bit-exact; `shift` offsets the three arrays off
    16-byte alignment (scalar-load path) and leaves a ragged tail."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows = cols = 1 << 13
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 17, degrees=G.powerlaw_degrees(rows, 1 << 17, cap=1 << 12))
    ri, ci, v = _coo(off, idx, val)
    n = ri.size - 5 if shift else ri.size   # ragged tail: not a multiple of 8
    xh = G.uniform_distribution_int(cols)
    want = np.zeros(rows, np.float64)
    np.add.at(want, ri[:n], v[:n].astype(np.float64) * xh[ci[:n]].astype(np.float64))
    def dev(a):
        buf = torch.zeros(a.size + shift, dtype=torch.from_numpy(a[:1]).dtype, device="cuda")
        buf[shift:] = torch.from_numpy(a).cuda()
        return buf[shift:shift + n]
    for tuned in (False, True):
        y = S.coo_spmv(rows, cols, dev(ri), dev(ci), dev(v), torch.from_numpy(xh).cuda(), tuned=tuned).cpu().numpy()
        assert np.array_equal(y.astype(np.float64), want), (shift, tuned)
    if not shift:
        assert np.array_equal(y, O.spmv_f32(off, idx, val, xh))


def test_plans_refuse_indices_outside_the_matrix():
    """loops_csc_plan_create_* / loops_coo_plan_create_*: a row index (COO: or a column index) outside the matrix must not be
    counted into a neighbouring row's offset or the temporaries: LOOPS_E_BADARG, nothing corrupted (a good plan still works)."""
    from loops_amd import spmv as S, _lib as L
    rows, cols = 4, 5
    good_r = torch.tensor([0, 1, 3, 3], dtype=torch.int32, device="cuda")
    good_c = torch.tensor([0, 4, 2, 3], dtype=torch.int32, device="cuda")
    vals = torch.ones(4, device="cuda")
    for bad_r, bad_c in (([0, 4, 3, 3], [0, 4, 2, 3]), ([0, -1, 3, 3], [0, 4, 2, 3]), ([0, 1, 3, 3], [0, 5, 2, 3])):
        with pytest.raises(L.LoopsError, match="BADARG"):
            S.COOPlan(rows, cols, torch.tensor(bad_r, dtype=torch.int32, device="cuda"),
                      torch.tensor(bad_c, dtype=torch.int32, device="cuda"), vals, measure=False)
    col_off = torch.tensor([0, 1, 1, 2, 3, 4], dtype=torch.int32, device="cuda")
    with pytest.raises(L.LoopsError, match="BADARG"):
        S.CSCPlan(rows, cols, col_off, torch.tensor([0, 9, 3, 1], dtype=torch.int32, device="cuda"), vals, measure=False)
    plan = S.COOPlan(rows, cols, good_r, good_c, vals, measure=False)
    y = plan.spmv(torch.arange(1, cols + 1, dtype=torch.float32, device="cuda"))
    assert y.cpu().tolist() == [1.0, 5.0, 0.0, 7.0]
    plan.close()
