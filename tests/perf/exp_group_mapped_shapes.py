"""group_mapped (heavy groups shared out) for several group sizes / tile shapes: rows per group = threads per workgroup is a launch
parameter of the schedule, not part of its contract.  Needs build/variants/libgm_shapes.so (tests/perf/gm_shapes.hip).
usage: exp_group_mapped_shapes.py [c2|rmat|host|band|uniform ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from loops_amd import generate as G
from oracle import oracle as O
L = C.CDLL(os.path.join(ROOT, "build", "variants", "libgm_shapes.so"))
SHAPES = ["gm_256x8", "gm_512x4", "gm_512x8", "gm_1024x4", "gm_128x8"]
for s in SHAPES:
    getattr(L, s).argtypes = [C.c_int] * 3 + [C.c_void_p] * 7
    getattr(L, s + "_bytes").restype = C.c_longlong


def ms(fn, iters=10):
    for _ in range(2): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


def matrix(name):
    if name == "rmat":
        off, idx, val = G.rmat_csr(23, 23, relabel="none"); return off, idx, val, 1 << 23, 1 << 23
    if name == "c2":
        off, idx, val = G.powerlaw_csr(1 << 20, 1 << 20, 1 << 24); return off, idx, val, 1 << 20, 1 << 20
    rows = cols = 7_414_866
    deg = G.powerlaw_degrees(rows, 194_109_311, native=True)
    window = {"host": G.HOST_BLOCKED, "band": 65536, "uniform": None}[name]
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, native=True, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
    return off, idx, val, rows, cols


for name in (sys.argv[1:] or ["c2", "rmat", "host", "band"]):
    off, idx, val, rows, cols = matrix(name)
    nnz = idx.size
    xh = G.uniform_distribution_int(cols)
    want = O.spmv_f32(off, idx, val, xh, omp=True)
    d = [torch.from_numpy(a).cuda() for a in (off, idx, val, xh)]
    y = torch.empty(rows, device="cuda")
    out = []
    for s in SHAPES:
        scratch = torch.zeros(int(getattr(L, s + "_bytes")(rows, nnz)) + 256, dtype=torch.uint8, device="cuda")
        fn = lambda: getattr(L, s)(rows, cols, nnz, *[a.data_ptr() for a in d], y.data_ptr(), scratch.data_ptr(), None)
        assert fn() == 0
        t = ms(fn, 50 if name == "c2" else 10)
        y.fill_(-1); fn()
        out.append("%s %.4f ms %s" % (s[3:], t, "ok" if np.array_equal(y.cpu().numpy(), want) else "WRONG"))
    print(name, " | ".join(out), flush=True)
