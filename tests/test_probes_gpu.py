"""The measurement instantiations of libloops_probes.so that the round-5 experiment records quote (profiles/r05_c3_lds_window_experiment.txt,
r05_c2_persistent_ordering_experiment.txt) compute the product's bits: the LDS window of x, the persistent workgroups with the
next tile's streams pipelined.  Not product code -- what is checked here is that the records compare like with like."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _matrix(window):
    from loops_amd import generate as G, spmv as S
    rows = cols = 1 << 16
    deg = G.powerlaw_degrees(rows, 1 << 20, cap=1 << 12)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    x = G.uniform_distribution_int(cols)
    return S.CSR.from_numpy(rows, cols, off, idx, val), (off, idx, val), x


@pytest.mark.parametrize("window", [None, 4096])
def test_windowed_gathers_and_pipelined_persistent_workgroups_are_bit_exact(window):
    from loops_amd import probes as PR
    from oracle import oracle as O
    csr, (off, idx, val), xh = _matrix(window)
    assert csr.nnzs % 4 == 0                      # (the pipelined form's stated precondition)
    ref = O.spmv_f32(off, idx, val, xh)
    x = torch.from_numpy(xh).cuda()
    y = torch.empty(csr.rows, device="cuda")
    names = PR.policies()
    windowed = [i for i, n in enumerate(names) if "window" in n]
    assert len(windowed) == 3
    run = PR.PolicyRunner(csr)
    for p in [0] + windowed:
        y.fill_(float("nan"))
        run.run(p, x, y)
        assert np.array_equal(y.cpu().numpy(), ref), names[p]
    pers = PR.PersistentRunner(csr)
    for groups in (1, 7, 64, 272):                # 272 merge tiles: one tile per workgroup at the top end
        for pipelined in (0, 1, 2):
            y.fill_(float("nan"))
            pers.run(pipelined, groups, x, y)
            assert np.array_equal(y.cpu().numpy(), ref), (groups, pipelined)
