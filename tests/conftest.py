import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# --------------------------------------------------------------------------------------------
# The hermetic matrix battery: same shapes the reference's SpMV battery exercises
# (unittests/test_spmv_battery.hxx:52-65: identity, diagonal, banded, block-diagonal, skewed,
# empty rows, random) restated as numpy recipes with committed seeds.  y for every entry is
# pinned by tests/golden/battery.npz, produced by the reference's own reference::spmv.
# --------------------------------------------------------------------------------------------
def _from_dense(a):
    a = np.asarray(a, np.float32)
    r, c = np.nonzero(a)
    off = np.zeros(a.shape[0] + 1, np.int32)
    np.add.at(off, r + 1, 1)
    return a.shape[0], a.shape[1], np.cumsum(off).astype(np.int32), c.astype(np.int32), a[r, c].astype(np.float32)


def _banded(n, lower, upper, rng):
    a = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in range(max(0, i - lower), min(n, i + upper + 1)):
            a[i, j] = rng.uniform(0.5, 1.5)
    return a


def battery():
    rng = np.random.default_rng(20260928)
    out = {}
    out["identity16"] = _from_dense(np.eye(16))
    out["diag16"] = _from_dense(np.diag(rng.uniform(0.5, 1.5, 16)))
    out["tridiag16"] = _from_dense(_banded(16, 1, 1, rng))
    out["banded32_3_4"] = _from_dense(_banded(32, 3, 4, rng))
    bd = np.zeros((8, 8), np.float32)
    for b in range(4):
        bd[2 * b:2 * b + 2, 2 * b:2 * b + 2] = rng.uniform(0.5, 1.5, (2, 2))
    out["block_diag4x2"] = _from_dense(bd)
    bd = np.zeros((9, 9), np.float32)
    for b in range(3):
        bd[3 * b:3 * b + 3, 3 * b:3 * b + 3] = rng.uniform(0.5, 1.5, (3, 3))
    out["block_diag3x3"] = _from_dense(bd)
    sk = np.zeros((20, 50), np.float32)
    sk[0, :] = rng.uniform(0.5, 1.5, 50)      # one heavy row, the rest light
    sk[7, ::3] = rng.uniform(0.5, 1.5, 17)
    for i in range(1, 20):
        sk[i, rng.choice(50, 2, replace=False)] = rng.uniform(0.5, 1.5, 2)
    out["skewed20x50"] = _from_dense(sk)
    er = (rng.random((20, 12)) < 0.3) * rng.uniform(0.5, 1.5, (20, 12))
    er[::4] = 0
    out["empty_rows20x12"] = _from_dense(er)
    out["random50"] = _from_dense((rng.random((50, 50)) < 0.05) * rng.uniform(-1.0, 1.0, (50, 50)))
    out["all_empty6"] = (6, 4, np.zeros(7, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    out["single_row"] = _from_dense(rng.uniform(0.5, 1.5, (1, 40)))
    out["single_col"] = _from_dense(rng.uniform(0.5, 1.5, (40, 1)))
    wide = (rng.random((300, 700)) < 0.02) * rng.uniform(-1, 1, (300, 700))
    wide[5, :] = rng.uniform(-1, 1, 700)       # a row longer than a 4x2 .. 128x7 merge tile
    wide[100:140] = 0                           # a run of 40 empty rows
    out["wide300x700"] = _from_dense(wide)
    return out


@pytest.fixture(scope="session")
def matrices():
    return battery()


@pytest.fixture(scope="session", autouse=True)
def _built_extension():
    """The HIP extension is built in-tree before any test runs (hipcc cross-compiles without a GPU;
    a no-op when libloops_amd.so is newer than its sources).  There is no fallback path to test:
    if this fails, every product call would raise LoopsError."""
    from loops_amd import _lib, generate
    _lib.build()
    generate.build_native()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
