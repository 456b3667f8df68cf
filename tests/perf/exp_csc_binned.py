"""One-shot CSC SpMV: binned products (kernels::launch_csc_binned) against the atomic kernel on the C2 matrix's transpose-free CSC
and on skewed row populations.  Needs build/variants/libcsc_binned.so (tests/perf/csc_binned.hip)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
from loops_amd import generate as G
L = C.CDLL(os.environ.get("CSC_LIB", os.path.join(ROOT, "build", "variants", "libcsc_binned.so")))
L.csc_binned_bytes.restype = C.c_longlong
vp = C.c_void_p


def ms(fn, iters=20):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


def csc_of(off, idx, val, rows, cols):
    m = sp.csr_matrix((val, idx, off), shape=(rows, cols)).tocsc()
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)


cases = {"c2": lambda: (G.powerlaw_csr(1 << 20, 1 << 20, 1 << 24), 1 << 20, 1 << 20),
         "rmat20": lambda: (G.rmat_csr(20, 16, relabel="none"), 1 << 20, 1 << 20),
         "small": lambda: (G.powerlaw_csr(5000, 7000, 60000), 5000, 7000),
         "one_hub_row": lambda: (G.csr_from_degrees(np.where(np.arange(1 << 16) == 7, 1 << 19, 4).astype(np.int64), 1 << 20, 1, 0, True, None), 1 << 16, 1 << 20)}
for name in (sys.argv[1:] or list(cases)):
    (off, idx, val), rows, cols = cases[name]()
    coff, ridx, cval = csc_of(off, idx, val, rows, cols)
    nnz = ridx.size
    xh = G.uniform_distribution_int(cols)
    want = (sp.csr_matrix((val.astype(np.float64), idx, off), shape=(rows, cols)) @ xh.astype(np.float64)).astype(np.float32)
    d = [torch.from_numpy(a).cuda() for a in (coff, ridx, cval, xh)]
    y1 = torch.full((rows,), 5.0, device="cuda"); y2 = torch.full((rows,), 5.0, device="cuda")
    scratch = torch.empty(int(L.csc_binned_bytes(rows, nnz)), dtype=torch.uint8, device="cuda")
    f1 = lambda: L.csc_binned(rows, cols, nnz, *[vp(a.data_ptr()) for a in d], vp(y1.data_ptr()), vp(scratch.data_ptr()), None)
    f2 = lambda: L.csc_atomic(rows, cols, nnz, *[vp(a.data_ptr()) for a in d], vp(y2.data_ptr()), None)
    assert f1() == 0 and f2() == 0
    t1, t2 = ms(f1), ms(f2)
    f1(); f2(); torch.cuda.synchronize()
    print("%-12s nnz %9d  binned %.3f ms exact=%s   atomic %.3f ms exact=%s" % (name, nnz, t1, bool(np.array_equal(y1.cpu().numpy(), want)), t2,
          bool(np.array_equal(y2.cpu().numpy(), want))), flush=True)
