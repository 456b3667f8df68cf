"""SURVEY 8(f) row 4 + BASELINE config C4: the other sparse formats' SpMV -- ours tuned, ours in the reference's
shape, and the REFERENCE'S OWN HIP build (oracle/_ref/libloops_ref_gpu.so) on the same GPU; every result
compared with the tuned CSR merge_path_flat result of the same matrix (exactly-summable inputs: bit-exact)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib

def ev(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

so = os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")
R = _lib.load_shared(so) if os.path.exists(so) else None
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
out = {}

def ref_format(fmt, rows, cols, off, idx, val, xh):
    if R is None: return None, None
    y = np.zeros(rows, np.float32); ms = C.c_float()
    rc = R.refgpu_format_spmv_f32(fmt, C.c_long(rows), C.c_long(cols), C.c_long(idx.size), p(off), p(idx), p(val), p(xh), p(y), 3, C.byref(ms))
    return (ms.value if rc == 0 else None), y

# ---- C2 as COO and as CSC
rows = cols = 1 << 20
off, idx, val = G.powerlaw_csr(rows, cols, 1 << 24)
off = off.astype(np.int32); idx = idx.astype(np.int32)
xh = G.uniform_distribution_int(cols); x = torch.from_numpy(xh).cuda()
csr = S.CSR.from_numpy(rows, cols, off, idx, val); want = S.spmv("merge_path_flat", csr, x)
ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(off))
d = {k: torch.from_numpy(v).cuda() for k, v in (("ri", ri), ("ci", idx), ("v", val))}
y = torch.empty(rows, device="cuda")
row = {}
for tuned in (True, False):
    t = ev(lambda: S.coo_spmv(rows, cols, d["ri"], d["ci"], d["v"], x, y, tuned=tuned))
    row["tuned" if tuned else "reference-shaped"] = {"ms": round(t, 4), "equal": bool(torch.equal(y, want))}
t, yr = ref_format(0, rows, cols, off, idx, val, xh)
row["reference build"] = {"ms": None if t is None else round(t, 4), "equal": None if yr is None else bool(np.array_equal(yr, want.cpu().numpy()))}
out["COO, C2 matrix"] = row; print("COO", row, file=sys.stderr, flush=True)
order = np.lexsort((ri, idx)); coff = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=cols))]).astype(np.int32)
dc = {k: torch.from_numpy(v).cuda() for k, v in (("off", coff), ("r", ri[order]), ("v", val[order]))}
row = {}
for tuned in (True, False):
    t = ev(lambda: S.csc_spmv(rows, cols, dc["off"], dc["r"], dc["v"], x, y, tuned=tuned))
    row["tuned" if tuned else "reference-shaped"] = {"ms": round(t, 4), "equal": bool(torch.equal(y, want))}
t, yr = ref_format(1, rows, cols, off, idx, val, xh)
row["reference build"] = {"ms": None if t is None else round(t, 4), "equal": None if yr is None else bool(np.array_equal(yr, want.cpu().numpy()))}
out["CSC, C2 matrix"] = row; print("CSC", row, file=sys.stderr, flush=True)
del csr, d, dc

# ---- ELL: 2^20 rows x exactly 16 nonzeros (pitch 16)
pitch = 16
rng = np.random.default_rng(0)
ind = np.sort(rng.integers(0, cols, size=(rows, pitch)).astype(np.int32), axis=1)
ev_ = (rng.integers(1, 9, size=(rows, pitch)) / 8.0).astype(np.float32)
eoff = (np.arange(rows + 1, dtype=np.int64) * pitch).astype(np.int32)
csr = S.CSR.from_numpy(rows, cols, eoff, ind.reshape(-1), ev_.reshape(-1)); want = S.spmv("merge_path_flat", csr, x)
di, dv = torch.from_numpy(ind).cuda(), torch.from_numpy(ev_).cuda()
row = {}
for tuned in (True, False):
    t = ev(lambda: S.ell_spmv(rows, cols, pitch, di, dv, x, y, tuned=tuned))
    row["tuned" if tuned else "reference-shaped"] = {"ms": round(t, 4), "equal": bool(torch.equal(y, want))}
t, yr = ref_format(2, rows, cols, eoff, ind.reshape(-1), ev_.reshape(-1), xh)
row["reference build"] = {"ms": None if t is None else round(t, 4), "equal": None if yr is None else bool(np.array_equal(yr, want.cpu().numpy()))}
out["ELL, 2^20 rows x pitch 16"] = row; print("ELL", row, file=sys.stderr, flush=True)
del csr, di, dv

# ---- C4: BCSR 4x4
nbr, per = 1 << 18, 16
boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
xb = G.uniform_distribution_int(nbr * 4); xbd = torch.from_numpy(xb).cuda()
b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
yb = torch.empty(nbr * 4, device="cuda")
row = {}
t = ev(lambda: S.bcsr_thread_mapped(b, xbd, yb, mfma=1), 30); wantb = yb.clone()
row["tuned (MFMA)"] = {"ms": round(t, 4), "equal": True}
t = ev(lambda: S.bcsr_thread_mapped(b, xbd, yb, mfma=0), 30)
row["reference-shaped"] = {"ms": round(t, 4), "equal": bool(torch.equal(yb, wantb))}
if R is not None:
    yr = np.zeros(nbr * 4, np.float32); ms = C.c_float()
    rc = R.refgpu_bcsr4x4_spmv_f32(C.c_long(nbr * 4), C.c_long(nbr * 4), C.c_long(nbr), C.c_long(nbr), C.c_long(bcols.size),
                                   p(boff), p(bcols), p(bvals), p(xb), p(yr), 5, C.byref(ms))
    row["reference build"] = {"ms": round(ms.value, 4) if rc == 0 else None, "equal": bool(np.array_equal(yr, wantb.cpu().numpy()))}
out["BCSR 4x4, C4"] = row; print("BCSR", row, file=sys.stderr, flush=True)
print(json.dumps(out))
