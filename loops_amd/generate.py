"""Synthetic inputs of the hot path (host side, numpy): the x-vector generator of the
reference's examples and the power-law CSR / uniform BCSR workloads named by BASELINE.json.

* ``uniform_distribution_int``  restates ``generate::random::uniform_distribution`` for INT
  bounds (reference include/loops/util/generate.hxx:33-79; every example calls it with
  (1, 10, seed 42): examples/spmv/merge_path.cu:33) -- vectorised; pinned against the
  reference build and the oracle in tests/test_generate.py.
* ``powerlaw_csr``  the C2 / C5 workload of SURVEY 8(d): clamped-Zipf row degrees
  (alpha = 0.8, cap 2^14), rows scattered by a fixed permutation, per-row distinct hashed
  columns sorted ascending, values k/8 and integer x so every row sum is exactly
  representable in fp32 (any summation order gives the same bits).  Counter-based: any row
  range can be generated independently (each rank of a multi-GPU run builds only its shard).
* ``uniform_bcsr``  the C4 workload: 4x4 blocks, a fixed number of blocks per block-row.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)

# ------------------------------------------------------------------------------ native builder
# The numpy functions below are the SPECIFICATION of the workloads; libloops_gen.so (csrc/loops_gen.cpp, host C++ +
# OpenMP) computes the same arrays bit for bit and is what the big configurations use (C3 stand-in: 194 M nonzeros,
# C5: 537 M) -- numpy needs minutes there.  tests/test_generate.py compares the two.  Every generator takes
# ``native=None`` (use the library when it is built), ``True`` (require it) or ``False`` (numpy).
_HERE = os.path.dirname(os.path.abspath(__file__))
GEN_LIB_PATH = os.path.join(_HERE, "libloops_gen.so")
GEN_SRC_PATH = os.path.join(_HERE, "csrc", "loops_gen.cpp")
_gen = None
_UNIFORM = -(1 << 63)


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (an OpenMP team larger
    than the quota is throttled to a crawl on the GPU boxes: 256 logical CPUs, 16 CPUs' worth of time)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    # one process per GPU (torchrun): the ranks of a node share the CPUs
    try:
        n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    return n


def build_native(force: bool = False) -> str:
    """g++ -O3 -fopenmp build of the native workload builder (host code only)."""
    if not force and os.path.exists(GEN_LIB_PATH) and os.path.getmtime(GEN_LIB_PATH) >= os.path.getmtime(GEN_SRC_PATH):
        return GEN_LIB_PATH
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-fopenmp", "-shared", "-fPIC", GEN_SRC_PATH, "-o", GEN_LIB_PATH])
    return GEN_LIB_PATH


def native(required: bool = False):
    """The loaded libloops_gen.so, or None when it has not been built (``required``: raise instead)."""
    global _gen
    if _gen is None and os.path.exists(GEN_LIB_PATH):
        L = C.CDLL(GEN_LIB_PATH)
        ll, vp = C.c_longlong, C.c_void_p
        L.loops_gen_degree_total.argtypes = [vp, ll, C.c_double, ll]
        L.loops_gen_degree_total.restype = ll
        L.loops_gen_degrees.argtypes = [vp, ll, C.c_double, ll, vp]
        L.loops_gen_degrees.restype = None
        L.loops_gen_perm_keys.argtypes = [C.c_ulonglong, ll, vp]
        L.loops_gen_perm_keys.restype = None
        L.loops_gen_csr_rows.argtypes = [vp, vp, ll, ll, C.c_ulonglong, ll, C.c_int, ll, vp, vp]
        L.loops_gen_csr_rows_hosts.argtypes = [vp, vp, ll, ll, C.c_ulonglong, ll, C.c_int, vp, ll, vp, vp]
        L.loops_gen_x_int.argtypes = [ll, ll, C.c_int, C.c_int, C.c_uint, vp]
        L.loops_gen_x_int.restype = None
        L.loops_gen_rmat_edges.argtypes = [C.c_int, ll, C.c_double, C.c_double, C.c_double, C.c_ulonglong, vp, vp]
        L.loops_gen_rmat_edges.restype = None
        L.loops_gen_set_threads.argtypes = [C.c_int]
        L.loops_gen_set_threads.restype = None
        L.loops_gen_set_threads(usable_cpus())
        _gen = L
    if _gen is None and required:
        raise RuntimeError(f"{GEN_LIB_PATH} not built (loops_amd.generate.build_native())")
    return _gen


def _use_native(flag):
    return native(required=True) if flag else (None if flag is False else native())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def splitmix64(z):
    """SplitMix64 finaliser, vectorised over uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (np.asarray(z, np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


# ------------------------------------------------------------------------------ x generator
def hash32(a):
    """generate.hxx:33-41 integer mix, vectorised (uint32 wrap-around)."""
    a = np.asarray(a, np.uint64)
    m = np.uint64(0xFFFFFFFF)
    a = ((a + np.uint64(0x7ED55D16)) + (a << np.uint64(12))) & m
    a = ((a ^ np.uint64(0xC761C23C)) ^ (a >> np.uint64(19))) & m
    a = ((a + np.uint64(0x165667B1)) + (a << np.uint64(5))) & m
    a = ((a + np.uint64(0xD3A2646C)) ^ (a << np.uint64(9))) & m
    a = ((a + np.uint64(0xFD7046C5)) + (a << np.uint64(3))) & m
    a = ((a ^ np.uint64(0xB55A4F09)) ^ (a >> np.uint64(16))) & m
    return a


def uniform_distribution_int(n, lo=1, hi=10, seed=42, dtype=np.float32, start=0, native=None):
    """x[i] for i in [start, start + n): minstd_rand seeded with hash(i) * seed, one draw, mapped
    through uniform_int_distribution<int>(lo, hi) (rocThrust semantics)."""
    L = _use_native(native) if n >= (1 << 16) or native else None
    if L is not None:
        x = np.empty(n, np.float32)
        L.loops_gen_x_int(int(start), int(n), int(lo), int(hi), int(seed) & 0xFFFFFFFF, _p(x))
        return x if dtype == np.float32 else x.astype(dtype)
    m = np.uint64(2147483647)
    i = np.arange(start, start + n, dtype=np.uint64)
    s = (hash32(i & np.uint64(0xFFFFFFFF)) * np.uint64(seed & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    s = s % m
    s[s == 0] = 1
    u = (np.uint64(48271) * s) % m
    r = (u - np.uint64(1)).astype(np.float64) / (1.0 + float(int(m) - 2))
    v = r * ((float(hi) + 1.0) - float(lo)) + float(lo)
    return v.astype(np.int64).astype(dtype)


# ------------------------------------------------------------------------------ power-law CSR
def powerlaw_degrees(rows, nnz, alpha=0.8, cap=1 << 14, perm_seed=0x5EED, native=None):
    """Row degrees d = clamp(floor(c / (rank + 1)^alpha), 1, cap) with c bisected so sum(d) == nnz,
    assigned to rows through the fixed permutation argsort(splitmix64(perm_seed ^ r))."""
    rank = np.arange(1, rows + 1, dtype=np.float64) ** (-alpha)
    L = _use_native(native)

    def total(c):
        if L is not None:
            return int(L.loops_gen_degree_total(_p(rank), rows, float(c), int(cap)))
        return int(np.clip(np.floor(c * rank), 1, cap).sum())

    lo, hi = 0.0, float(cap) * rows
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if total(mid) < nnz:
            lo = mid
        else:
            hi = mid
    if L is not None:
        d = np.empty(rows, np.int64)
        L.loops_gen_degrees(_p(rank), rows, float(hi), int(cap), _p(d))
    else:
        d = np.clip(np.floor(hi * rank), 1, cap).astype(np.int64)
    diff = int(d.sum()) - nnz  # hi gives total >= nnz: shave the residual off the largest uncapped rows
    if diff:
        idx = np.flatnonzero((d < cap) & (d > 1)) if diff > 0 else np.flatnonzero(d < cap)
        assert idx.size >= abs(diff), "cannot fix degree residual"
        d[idx[: abs(diff)]] -= np.sign(diff)
    assert int(d.sum()) == nnz
    if L is not None:
        keys = np.empty(rows, np.uint64)
        L.loops_gen_perm_keys(int(perm_seed), rows, _p(keys))
    else:
        keys = splitmix64(np.uint64(perm_seed) ^ np.arange(rows, dtype=np.uint64))
    perm = np.argsort(keys, kind="stable")
    out = np.empty(rows, np.int64)
    out[perm] = d  # rank k's degree lands on row perm[k]
    return out


HOST_BLOCKED = -4  # `window` value: columns drawn "host-blocked" (needs `hosts`, see host_blocks)


def host_blocks(n, seed=11, smin=256, smax=1 << 17, alpha=1.1):
    """Boundaries of consecutive-id blocks ("hosts") covering [0, n) with power-law sizes -- how a crawl-ordered web graph
    such as LAW/indochina-2004 is laid out: pages of one host get consecutive ids and most links stay inside the host.
    size_i = clamp(floor(smin * u_i^(-1 / alpha)), smin, smax), u_i hashed from (seed, i); the last block is cut at n.
    Returns int64 boundaries b[0] = 0 < ... < b[H] = n."""
    out = [0]
    i = 0
    while out[-1] < n:
        m = max(1024, (n - out[-1]) // smin // 4 + 1)  # a batch of hashed sizes at a time
        u = ((splitmix64(np.arange(i, i + m, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x51ED27)) >> np.uint64(11)).astype(np.float64)
             + 1.0) / float(1 << 53)
        size = np.clip(np.floor(smin * u ** (-1.0 / alpha)), smin, smax).astype(np.int64)
        ends = out[-1] + np.cumsum(size)
        keep = int(np.searchsorted(ends, n, "left")) + 1
        out.extend(int(e) for e in ends[:keep])
        i += m
    out = np.asarray(out, np.int64)
    out = out[out < n]
    return np.concatenate([out, [n]]).astype(np.int64)


def _hash_cols(seed, rows_abs, k, attempt, cols, window=None, deg=None, hosts=None):
    h = splitmix64(splitmix64(np.uint64(seed) + rows_abs.astype(np.uint64)) + k.astype(np.uint64)
                   + (np.uint64(attempt) << np.uint64(40)))
    if window is None:
        return (h % np.uint64(cols)).astype(np.int64)
    if window == HOST_BLOCKED:
        # 14 of every 16 links stay inside the row's host (when the host can hold twice the row's degree), the rest go
        # anywhere: the locality class of a crawl-ordered web graph (host_blocks)
        ids = np.minimum(rows_abs, cols - 1)
        j = np.searchsorted(hosts, ids, "right") - 1
        hb, he = hosts[j], hosts[j + 1]
        size = he - hb
        local = ((h >> np.uint64(60)) < np.uint64(14)) & (size >= 2 * deg)
        inside = hb + ((h & np.uint64(0xFFFFFFFFFFFF)) % np.maximum(size, 1).astype(np.uint64)).astype(np.int64)
        anywhere = (splitmix64(h) % np.uint64(cols)).astype(np.int64)
        return np.where(local, inside, anywhere)
    if window <= -2:  # power-law COLUMN popularity too (scale-free in both dimensions, like R-MAT graphs):
        # column rank = cols * u^5 (Zipf-like, alpha = 0.8); -2: popular columns scattered by a fixed
        # multiplicative permutation, -3: popular columns adjacent (labels sorted by popularity)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        rank = np.minimum((np.float64(cols) * u ** 5).astype(np.int64), np.int64(cols - 1))
        if window == -3:
            return rank
        return (rank * np.int64(2654435761)) % np.int64(cols)  # odd multiplier: a permutation of [0, 2^k)
    if window < 0:  # "runs": the row's columns are consecutive from a hashed start (perfectly coalesced gather)
        start = splitmix64(np.uint64(seed) * np.uint64(31) + rows_abs.astype(np.uint64)) % np.uint64(cols)
        return ((start.astype(np.int64) + k) % np.int64(cols))
    # locality variant: columns fall in a window centred on the diagonal, at least 4x the row's
    # degree wide (so distinct columns always exist) and never wider than the matrix
    w = np.minimum(np.maximum(np.int64(window), 4 * deg), np.int64(cols)).astype(np.uint64)
    off = (h % w).astype(np.int64) - (w // np.uint64(2)).astype(np.int64)
    return (rows_abs + off) % np.int64(cols)


def csr_from_degrees(degrees, cols, seed=1, row_begin=0, exact=True, window=None, native=None, hosts=None):
    """Rows [row_begin, row_begin + len(degrees)) of the hashed matrix: per-row distinct columns
    sorted ascending; values k/8 (exact) or U[0.5, 1.5) (realistic).  `window` = None draws
    columns uniformly over [0, cols); an integer draws them from a band of that many
    columns around the diagonal (web-graph-like locality of the x gather)."""
    nrows = degrees.size
    offsets = np.zeros(nrows + 1, np.int64)
    np.cumsum(degrees, out=offsets[1:])
    nnz = int(offsets[-1])
    assert int(degrees.max(initial=0)) <= cols
    assert nnz < (1 << 31), "int32 offsets"
    if window == HOST_BLOCKED:
        assert hosts is not None and hosts[0] == 0 and hosts[-1] == cols, "window = HOST_BLOCKED needs hosts = host_blocks(cols)"
        hosts = np.ascontiguousarray(hosts, np.int64)
    L = _use_native(native) if (window is None or window in (-1, HOST_BLOCKED) or window > 0) else None
    if L is not None and window == HOST_BLOCKED:
        deg64 = np.ascontiguousarray(degrees, np.int64)
        indices = np.empty(nnz, np.int32)
        values = np.empty(nnz, np.float32)
        rc = L.loops_gen_csr_rows_hosts(_p(deg64), _p(offsets), nrows, int(cols), int(seed), int(row_begin), int(bool(exact)),
                                        _p(hosts), int(hosts.size - 1), _p(indices), _p(values))
        assert rc == 0, "a row cannot hold that many distinct columns"
        return offsets.astype(np.int32), indices, values
    if L is not None:
        deg64 = np.ascontiguousarray(degrees, np.int64)
        indices = np.empty(nnz, np.int32)
        values = np.empty(nnz, np.float32)
        rc = L.loops_gen_csr_rows(_p(deg64), _p(offsets), nrows, int(cols), int(seed), int(row_begin), int(bool(exact)),
                                  _UNIFORM if window is None else int(window), _p(indices), _p(values))
        assert rc == 0, "a row cannot hold that many distinct columns"
        return offsets.astype(np.int32), indices, values
    rloc = np.repeat(np.arange(nrows, dtype=np.int64), degrees)
    k = np.arange(nnz, dtype=np.int64) - np.repeat(offsets[:-1], degrees)
    rabs = rloc + row_begin
    dk = np.repeat(degrees.astype(np.int64), degrees) if window is not None else None
    col = _hash_cols(seed, rabs, k, 0, cols, window, dk, hosts)
    attempt = 0
    while True:
        key = (rloc.astype(np.uint64) << np.uint64(32)) | col.astype(np.uint64)
        order = np.argsort(key, kind="stable")
        sk = key[order]
        dup = np.flatnonzero(sk[1:] == sk[:-1]) + 1
        if dup.size == 0:
            break
        attempt += 1
        bad = order[dup]
        col[bad] = _hash_cols(seed, rabs[bad], k[bad], attempt, cols, window, None if dk is None else dk[bad], hosts)
    indices = (sk & np.uint64(0xFFFFFFFF)).astype(np.int32)  # sorted by (row, col)
    rsorted = rloc  # rows are already grouped: sorting by (row, col) keeps row order
    vh = splitmix64((rsorted + row_begin).astype(np.uint64) * np.uint64(0x100000001B3) + indices.astype(np.uint64)
                    + np.uint64(seed) * np.uint64(7919))
    if exact:
        values = (((vh >> np.uint64(33)) % np.uint64(8)) + np.uint64(1)).astype(np.float32) / np.float32(8)
    else:
        values = (0.5 + (vh >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)
    return offsets.astype(np.int32), indices, values


def powerlaw_csr(rows, cols, nnz, alpha=0.8, cap=1 << 14, seed=1, row_begin=0, row_end=None, exact=True,
                 degrees=None, window=None):
    """(offsets, indices, values) of rows [row_begin, row_end) of the power-law matrix
    (offsets rebased to 0, global column ids)."""
    if degrees is None:
        degrees = powerlaw_degrees(rows, nnz, alpha, min(cap, cols))
    row_end = rows if row_end is None else row_end
    return csr_from_degrees(degrees[row_begin:row_end], cols, seed, row_begin, exact, window)


def realistic_x(n, seed=7, start=0):
    h = splitmix64(np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9))
    return (0.5 + (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


# ------------------------------------------------------------------------------ R-MAT (Graph500-like graphs)
def rmat_edges(scale, nedges, a=0.57, b=0.19, c=0.19, seed=1, native=None):
    """Directed R-MAT / Kronecker edges over 2^scale vertices (Graph500: a, b, c = 0.57, 0.19, 0.19, d = 1 - a - b - c): edge e
    draws one hashed u per level -- u = (splitmix64(splitmix64(seed) + 64 e + level) >> 11) 2^-53 -- and descends into quadrant
    (row bit, column bit) = (0, 0) | (0, 1) | (1, 0) | (1, 1) for u < a | < a + b | < a + b + c | else; level 0 sets the most
    significant bit.  Counter-based (any range of edges independently); low vertex ids are the hubs.  -> (row, col) int32."""
    lib = _use_native(native)
    if lib is not None:
        row, col = np.empty(nedges, np.int32), np.empty(nedges, np.int32)
        lib.loops_gen_rmat_edges(int(scale), int(nedges), float(a), float(b), float(c), int(seed), row.ctypes.data, col.ctypes.data)
        return row, col
    base = splitmix64(np.array([seed], np.uint64))[0]
    e = np.arange(nedges, dtype=np.uint64) * np.uint64(64)
    row, col = np.zeros(nedges, np.int64), np.zeros(nedges, np.int64)
    for level in range(scale):
        u = (splitmix64(base + e + np.uint64(level)) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        rb = (u >= a + b).astype(np.int64)
        cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.int64)
        row, col = (row << 1) | rb, (col << 1) | cb
    return row.astype(np.int32), col.astype(np.int32)


def rmat_csr(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=1, relabel="random", native=None):
    """The adjacency matrix of an R-MAT graph as a CSR (2^scale rows and columns, edge_factor * 2^scale nonzeros, multi-edges kept
    as separate nonzeros, columns sorted inside a row, values k/8: exactly summable with integer x).  ``relabel``:
    "none"    the generator's own ids: hubs at the low ids, the recursive quadrant structure shows as locality;
    "random"  a hashed permutation of the ids (what Graph500 prescribes: no locality left);
    "degree"  ids by descending out-degree (a locality-restoring order: hub rows first, hub columns adjacent).
    -> (offsets int32, indices int32, values float32)"""
    n, nedges = 1 << scale, edge_factor << scale
    row, col = rmat_edges(scale, nedges, a, b, c, seed, native)
    row, col = row.astype(np.int64), col.astype(np.int64)
    if relabel == "random":
        perm = np.argsort(splitmix64(np.arange(n, dtype=np.uint64) ^ np.uint64(seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)), kind="stable")
        new_id = np.empty(n, np.int64)
        new_id[perm] = np.arange(n)
        row, col = new_id[row], new_id[col]
    elif relabel == "degree":
        deg = np.bincount(row, minlength=n) + np.bincount(col, minlength=n)
        order = np.argsort(-deg, kind="stable")
        new_id = np.empty(n, np.int64)
        new_id[order] = np.arange(n)
        row, col = new_id[row], new_id[col]
    elif relabel != "none":
        raise ValueError("relabel: none | random | degree")
    key = (row << 32) | col
    key.sort()
    row, col = key >> 32, key & 0xFFFFFFFF
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=offsets[1:])
    vh = splitmix64(np.arange(nedges, dtype=np.uint64) + np.uint64(seed) * np.uint64(7919))
    values = (((vh >> np.uint64(33)) % np.uint64(8)) + np.uint64(1)).astype(np.float32) / np.float32(8)
    return offsets.astype(np.int32), col.astype(np.int32), values


# ------------------------------------------------------------------------------ BCSR
def uniform_bcsr(num_block_rows, num_block_cols, blocks_per_row, R=4, C=4, seed=3):
    """C4: every block-row holds `blocks_per_row` blocks at distinct hashed block columns (sorted),
    block cells k/8."""
    deg = np.full(num_block_rows, blocks_per_row, np.int64)
    boff, bcols, _ = csr_from_degrees(deg, num_block_cols, seed)
    nb = bcols.size
    cell = np.arange(nb * R * C, dtype=np.uint64)
    vh = splitmix64(cell + np.uint64(seed) * np.uint64(104729))
    vals = (((vh >> np.uint64(33)) % np.uint64(8)) + np.uint64(1)).astype(np.float32) / np.float32(8)
    return boff, bcols, vals


# ------------------------------------------------------------------------------ small fixtures
def random_csr(rows, cols, density, seed, empty_every=0, values="uniform"):
    """Small hermetic matrices for parity tests (numpy Generator, committed seeds)."""
    rng = np.random.default_rng(seed)
    mask = rng.random((rows, cols)) < density
    if empty_every:
        mask[::empty_every] = False
    r, c = np.nonzero(mask)
    offsets = np.zeros(rows + 1, np.int32)
    np.add.at(offsets, r + 1, 1)
    offsets = np.cumsum(offsets).astype(np.int32)
    if values == "uniform":
        v = (rng.random(r.size) * 2 - 1).astype(np.float32)
    else:
        v = (rng.integers(1, 9, r.size) / 8).astype(np.float32)
    return offsets, c.astype(np.int32), v
