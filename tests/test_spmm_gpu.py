"""GPU parity of the CSR SpMM path through the C ABI, against the CPU oracle's
restatement of algorithms::spmm::thread_mapped (spmm/thread_mapped.cuh:38-51).  Integer B and
dyadic values are exactly summable in fp32 -> BIT-EXACT in any summation order; real-valued
inputs are held to 1e-6 relative to the row's L1 mass."""
import numpy as np
import pytest

from conftest import battery

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
WIDTHS = [1, 3, 8, 10, 16, 17, 32, 33, 64, 65, 100, 130]


def _dev(off, idx, val, rows, cols):
    from loops_amd import spmv as S
    return S.CSR.from_numpy(rows, cols, off, idx, val)


def _dyadic(val):
    """Values k/8, k in 1..8, derived from the battery's values (sign kept)."""
    k = (np.abs(val.astype(np.float64)) * 1e3).astype(np.int64) % 8 + 1
    return (np.sign(val) + (val == 0)).astype(np.float32) * (k / 8.0).astype(np.float32)


@pytest.mark.parametrize("schedule", ["merge_path_flat", "thread_mapped"])
def test_battery_bit_exact(schedule):
    from loops_amd import spmv as S
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    for name, (r, c, off, idx, val) in battery().items():
        v = _dyadic(val)
        csr = _dev(off, idx, v, r, c)
        for n in WIDTHS:
            B = rng.integers(1, 11, size=(c, n)).astype(np.float32)
            Cd = torch.full((r, n), 7.0, device="cuda")  # no zero-fill needed
            S.spmm(csr, torch.from_numpy(B).cuda(), Cd, schedule=schedule)
            assert np.array_equal(Cd.cpu().numpy(), O.spmm(off, idx, v, B)), (schedule, name, n)


def test_matches_spmv_column_by_column():
    """Column j of C equals the tuned SpMV with x = B[:, j] (same exactly-summable inputs)."""
    from loops_amd import spmv as S, generate as G
    rows = cols = 1 << 12
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 16, degrees=G.powerlaw_degrees(rows, 1 << 16, cap=1 << 11))
    csr = _dev(off, idx, val, rows, cols)
    rng = np.random.default_rng(3)
    B = torch.from_numpy(rng.integers(1, 11, size=(cols, 12)).astype(np.float32)).cuda()
    Cd = S.spmm(csr, B)
    for j in range(B.shape[1]):
        y = S.spmv("merge_path_flat", csr, B[:, j].contiguous())
        assert torch.equal(Cd[:, j], y), j


@pytest.mark.parametrize("n", [10, 32, 64, 96])
def test_powerlaw_rows_spanning_tiles(n):
    """Rows far longer than a merge tile (2^12 nonzeros vs 2048-item tiles): n-wide carry-outs and
    the fix-up kernel; with and without a held plan; f32 and f64."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows = cols = 1 << 13
    deg = G.powerlaw_degrees(rows, 1 << 17, cap=1 << 12)
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 17, degrees=deg)
    rng = np.random.default_rng(n)
    B = rng.integers(1, 11, size=(cols, n)).astype(np.float32)
    ref = O.spmm(off, idx, val, B)
    csr = _dev(off, idx, val, rows, cols)
    Bd = torch.from_numpy(B).cuda()
    assert np.array_equal(S.spmm(csr, Bd).cpu().numpy(), ref)
    plan = S.MergePathPlan(csr)
    for _ in range(2):  # second call reuses the carry-out allocation
        assert np.array_equal(S.spmm(csr, Bd, plan=plan).cpu().numpy(), ref)
    csr64 = _dev(off, idx, val.astype(np.float64), rows, cols)
    C64 = S.spmm(csr64, Bd.double()).cpu().numpy()
    assert np.array_equal(C64, O.spmm(off, idx, val.astype(np.float64), B.astype(np.float64)))


def test_real_values_within_tolerance():
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows = cols = 1 << 12
    off, idx, _ = G.powerlaw_csr(rows, cols, 1 << 16, degrees=G.powerlaw_degrees(rows, 1 << 16, cap=1 << 11))
    rng = np.random.default_rng(11)
    val = rng.uniform(0.5, 1.5, size=idx.size).astype(np.float32)
    B = rng.uniform(0.5, 1.5, size=(cols, 24)).astype(np.float32)
    ref = O.spmm(off, idx, val.astype(np.float64), B.astype(np.float64))  # f64 reference
    l1 = O.spmm(off, idx, np.abs(val).astype(np.float64), np.abs(B).astype(np.float64))
    got = S.spmm(_dev(off, idx, val, rows, cols), torch.from_numpy(B).cuda()).cpu().numpy()
    assert np.all(np.abs(got - ref) <= 2e-6 * l1 + 1e-30)


@pytest.mark.parametrize("shift", [1, 2])
def test_unaligned_bases_fall_back_to_narrower_accesses(shift):
    """B / C bases that are only 4- or 8-byte aligned: the launcher must drop to 4- / 8-byte accesses."""
    from loops_amd import spmv as S, generate as G
    from oracle import oracle as O
    rows = cols = 1 << 11
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 15, degrees=G.powerlaw_degrees(rows, 1 << 15, cap=1 << 10))
    csr = _dev(off, idx, val, rows, cols)
    rng = np.random.default_rng(shift)
    for n in (8, 16, 64):
        Bh = rng.integers(1, 11, size=(cols, n)).astype(np.float32)
        Bbuf = torch.zeros(cols * n + shift, device="cuda")
        Bbuf[shift:] = torch.from_numpy(Bh).cuda().flatten()
        Cbuf = torch.zeros(rows * n + shift, device="cuda")
        S.spmm(csr, Bbuf[shift:].view(cols, n), Cbuf[shift:].view(rows, n))
        assert np.array_equal(Cbuf[shift:].view(rows, n).cpu().numpy(), O.spmm(off, idx, val, Bh)), (shift, n)
        assert Cbuf[:shift].abs().sum().item() == 0


def test_empty_and_degenerate():
    from loops_amd import spmv as S
    z = np.zeros(0, np.int32)
    # no nonzeros at all: C must come back all zero
    csr = _dev(np.zeros(6, np.int32), z, np.zeros(0, np.float32), 5, 4)
    Cd = torch.full((5, 9), 3.0, device="cuda")
    S.spmm(csr, torch.ones((4, 9), device="cuda"), Cd)
    assert torch.count_nonzero(Cd).item() == 0
    # n == 0 and rows == 0 are no-ops
    S.spmm(csr, torch.ones((4, 0), device="cuda"))
    S.spmm(_dev(np.zeros(1, np.int32), z, np.zeros(0, np.float32), 0, 4), torch.ones((4, 3), device="cuda"))
    # unsupported schedule for SpMM is refused, not silently rerouted
    from loops_amd import _lib
    with pytest.raises(_lib.LoopsError):
        S.spmm(csr, torch.ones((4, 9), device="cuda"), schedule="group_mapped")
