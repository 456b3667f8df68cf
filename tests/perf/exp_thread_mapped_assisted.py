"""thread_mapped with wavefront-assisted long rows (kernels::thread_mapped_assisted_spmv) against the batched kernel and the
reference-shaped loop: bits and time.  Needs build/variants/libtm_assist.so (built from tests/perf/tm_assist.hip: command in its header;
since the kernel became the tuned launch, "batched" = tm_old shows the assisted kernel too -- the batched figures are in the record)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from loops_amd import generate as G
L = C.CDLL(os.environ.get("TM_LIB", os.path.join(ROOT, "build", "variants", "libtm_assist.so")))
vp = C.c_void_p
for f in ("tm_old", "tm_new", "tm_ref", "tm_new_f64"):
    getattr(L, f).argtypes = [C.c_int] * 3 + [vp] * 6


def ms(fn, iters=10):
    for _ in range(2): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


cases = {"c2": lambda: G.powerlaw_csr(1 << 20, 1 << 20, 1 << 24, exact=False),
         "rows16_band": lambda: G.csr_from_degrees(np.full(1 << 20, 16, np.int64), 1 << 20, 1, 0, False, 64),
         "rows16_uniform": lambda: G.csr_from_degrees(np.full(1 << 20, 16, np.int64), 1 << 20, 1, 0, False, None),
         "rows200": lambda: G.csr_from_degrees(np.full(1 << 16, 200, np.int64), 1 << 20, 1, 0, False, None),
         "medium16": lambda: G.csr_from_degrees(np.where(np.arange(1 << 18) % 4 == 1, 400, 6).astype(np.int64), 1 << 20, 1, 0, False, None),
         "one300": lambda: G.csr_from_degrees(np.where(np.arange(1 << 19) % 64 == 9, 300, 12).astype(np.int64), 1 << 20, 1, 0, False, None),
         "one_huge": lambda: G.csr_from_degrees(np.where(np.arange(64) == 3, 1 << 19, 5).astype(np.int64), 1 << 20, 1, 0, False, None),
         "mixed": lambda: G.csr_from_degrees(np.where(np.arange(1 << 18) % 64 == 5, 900, 7).astype(np.int64), 1 << 20, 1, 0, False, None)}
for name in (sys.argv[1:] or list(cases)):
    off, idx, val = cases[name]()
    rows, cols, nnz = off.size - 1, 1 << 20, idx.size
    xh = (np.random.default_rng(1).random(cols) + 0.5).astype(np.float32)
    d = [torch.from_numpy(a).cuda() for a in (off, idx, val, xh)]
    ys = {k: torch.full((rows,), -1.0, device="cuda") for k in ("tm_old", "tm_new", "tm_ref")}
    t = {}
    for k in ys:
        fn = lambda k=k: getattr(L, k)(rows, cols, nnz, *[a.data_ptr() for a in d], ys[k].data_ptr(), None)
        assert fn() == 0
        t[k] = ms(fn, 5 if k == "tm_ref" else 20)
    d64 = [d[0], d[1], d[2].double(), d[3].double()]
    y64 = torch.empty(rows, dtype=torch.float64, device="cuda")
    f64 = lambda: L.tm_new_f64(rows, cols, nnz, *[a.data_ptr() for a in d64], y64.data_ptr(), None)
    f64(); t64 = ms(f64)
    prod = val.astype(np.float64) * xh.astype(np.float64)[idx]
    want = np.add.reduceat(np.concatenate([prod, [0.0]]), np.minimum(off[:-1], prod.size)); want[np.diff(off) == 0] = 0
    print("%-15s ref-shaped %.3f ms  batched %.3f ms  assisted %.3f ms (f64 %.3f)   assisted==ref bits %s  batched==ref %s  f64 max rel err %.1e" % (
        name, t["tm_ref"], t["tm_old"], t["tm_new"], t64, bool(torch.equal(ys["tm_new"], ys["tm_ref"])), bool(torch.equal(ys["tm_old"], ys["tm_ref"])),
        float(np.max(np.abs(y64.cpu().numpy() - want) / np.maximum(np.abs(want), 1e-30)))), flush=True)
