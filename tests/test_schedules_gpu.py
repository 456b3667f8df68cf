"""Bit-exact tile/atom index assignment: what the device schedules hand out vs the oracle."""
import numpy as np
import pytest

from conftest import battery

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cases():
    from loops_amd import generate as G
    cases = dict(battery())
    rows = cols = 1 << 12
    deg = G.powerlaw_degrees(rows, 1 << 16, cap=1 << 11)
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 16, degrees=deg)
    cases["powerlaw4096"] = (rows, cols, off, idx, val)
    return cases


@pytest.mark.parametrize("tile", ["256x8", "128x7", "4x2", "256x7", "512x8", "256x16"])
def test_merge_path_coordinates_and_thread_assignment(tile):
    from loops_amd import spmv as S, _lib
    from oracle import oracle as O
    _, tpb, ipt = _lib.TILES[tile]
    for name, (r, c, off, idx, val) in _cases().items():
        csr = S.CSR.from_numpy(r, c, off, idx, val)
        plan = S.MergePathPlan(csr, tile)
        assert np.array_equal(plan.coords(), O.merge_path_coords(off, tpb, ipt)), (name, tile)
        want = O.merge_path_assign(off, tpb, ipt)
        for use_plan in (True, False):  # precomputed table vs in-kernel two-lane search
            got = S.dump_merge_path(csr, tile, use_plan)
            for a, b, what in zip(got, want, ("thread_start", "owner", "row", "visits")):
                assert np.array_equal(a, b), (name, tile, use_plan, what)


def test_work_oriented_assignment():
    from loops_amd import spmv as S
    from oracle import oracle as O
    grid = S.work_oriented_grid()
    assert grid > 0
    for name, (r, c, off, idx, val) in _cases().items():
        csr = S.CSR.from_numpy(r, c, off, idx, val)
        for g in (1, 3, grid):
            got = S.dump_work_oriented(csr, g)
            want = O.work_oriented_assign(off, g * 256)
            for a, b, what in zip(got, want, ("thread_map", "owner", "row", "visits")):
                assert np.array_equal(a, b), (name, g, what)


@pytest.mark.parametrize("group", [256, 64, 16])
def test_group_mapped_assignment(group):
    from loops_amd import spmv as S
    from oracle import oracle as O
    for name, (r, c, off, idx, val) in _cases().items():
        csr = S.CSR.from_numpy(r, c, off, idx, val)
        got = S.dump_group_mapped(csr, group)
        want = O.group_mapped_assign(off, group)
        for a, b, what in zip(got, want, ("owner", "row", "visits")):
            assert np.array_equal(a, b), (name, group, what)
