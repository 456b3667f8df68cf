"""The numpy specification of the row-band layout (tests/rowband_spec.py) checked on the CPU: every nonzero appears exactly once,
the layout's own product (fp32 products, fp64 sums, hub replicas folded) equals the oracle's on the battery and on a matrix
with hubs and several column blocks, the chunk lists tile every band exactly for any target count.  The device builder is
compared with this specification array for array in tests/test_rowband_gpu.py."""
import numpy as np

from conftest import battery, load_golden
import rowband_spec as spec


def _csr_product(off, idx, val, x):
    prod = val.astype(np.float32) * x.astype(np.float32)[idx]
    y = np.zeros(off.size - 1)
    np.add.at(y, np.repeat(np.arange(off.size - 1), np.diff(off)), prod.astype(np.float64))
    return y


def test_layout_holds_every_nonzero_once_and_reproduces_the_product_on_the_battery():
    g = load_golden("battery.npz")
    for name, (r, c, off, idx, val) in battery().items():
        if r == 0:
            continue
        for H in (64, 256):
            v, r16, d8, perm, base, bs, hubs = spec.layout(off, idx, val, r, c, H)
            real = perm >= 0
            assert real.sum() == idx.size and np.array_equal(np.sort(perm[real]), np.arange(idx.size)), name
            assert v.size % spec.STEP == 0 and bs[0] == 0 and bs[-1] * spec.STEP == v.size and base.shape == (v.size // spec.STEP, 4)
            assert np.all(v[~real] == 0) and np.all(r16[~real] == H)
            x = g[f"{name}.x_int"].astype(np.float32)
            assert np.array_equal(spec.product(v, r16, d8, base, bs, hubs, H, r, x), _csr_product(off, idx, val, x)), (name, H)


def test_hubs_gap_pads_and_chunk_lists():
    from loops_amd import generate as G
    rows, cols = 9_001, 150_001
    deg = G.powerlaw_degrees(rows, 1 << 18, cap=1 << 12)
    deg[::7] = 0
    off, idx, val = G.csr_from_degrees(deg, cols, 1)
    H = 2048
    v, r16, d8, perm, base, bs, hubs = spec.layout(off, idx, val, rows, cols, H)
    hubidx, table = spec.hub_table(off, rows, H)
    assert np.array_equal(table, hubs) and hubs[:, 0].max() > 0 and hubs[:, 0].max() <= spec.MAX_HUBS
    assert int(r16.max()) <= H + spec.MAX_HUBS * spec.HUB_REPLICAS              # row codes fit the LDS words of a workgroup
    assert base.min() >= 0 and base.max() < cols
    x = G.uniform_distribution_int(cols)
    assert np.array_equal(spec.product(v, r16, d8, base, bs, hubs, H, rows, x), _csr_product(off, idx, val, x))
    # a sparse band: column gaps far beyond 255 are bridged by padding slots of delta 255 that contribute nothing
    o2, i2, v2 = G.csr_from_degrees(np.full(300, 3, np.int64), 3_000_000, 1)
    a = spec.layout(o2, i2, v2, 300, 3_000_000, 256)
    assert (a[3] < 0).sum() > 5 * i2.size and (a[2][a[3] < 0] == 255).sum() > i2.size
    x2 = G.uniform_distribution_int(3_000_000)
    assert np.array_equal(spec.product(a[0], a[1], a[2], a[4], a[5], a[6], 256, 300, x2), _csr_product(o2, i2, v2, x2))
    B = bs.size - 1
    for target in (1, B, 7, 64, 1000, 10 ** 6):
        ch, mu = spec.chunk_list(bs, target)
        assert ch.shape[0] >= B
        for b in range(B):
            mine = ch[ch[:, 0] == b]
            assert mine[0, 1] == bs[b] and mine[-1, 2] == bs[b + 1] and np.array_equal(mine[1:, 1], mine[:-1, 2])   # tiles the band
            assert np.all(mine[:, 2] > mine[:, 1]) or bs[b] == bs[b + 1]
            if mine.shape[0] > 1:
                first, count = mu[mu[:, 0] == b][0, 1:]
                assert count == mine.shape[0] and np.array_equal(mine[:, 3], first + np.arange(count))
            else:
                assert mine[0, 3] == -1
        # list order: by piece number first, bands ascending inside a piece number
        piece = np.array([int(np.sum((ch[:i, 0] == ch[i, 0]))) for i in range(ch.shape[0])])
        assert np.all(np.diff(piece) >= 0) and all(np.all(np.diff(ch[piece == p, 0]) > 0) for p in np.unique(piece))
        slots = ch[ch[:, 3] >= 0, 3]
        assert np.array_equal(np.sort(slots), np.arange(slots.size))
        if target >= B:
            assert ch.shape[0] <= max(target, B) + B
