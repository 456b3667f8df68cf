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
import time
for copy in (False, True):   # the held CSC plan: storage transposed once; with / without a re-ordered copy
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cp = S.CSCPlan(rows, cols, dc["off"], dc["r"], dc["v"], allow_copy=copy, measure=True)
    torch.cuda.synchronize(); build = (time.perf_counter() - t0) * 1e3
    t = ev(lambda: cp.spmv(x, y))
    row["held plan" + (", copy allowed" if copy else " (CSR copy only)")] = {"ms": round(t, 4), "layout": cp.layout, "build_ms": round(build, 1), "equal": bool(torch.equal(y, want))}
    cp.close()
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
# algorithms::spmv::ell_merge_path: ours on the fused engine vs the reference's atomic kernel (format 4)
t = ev(lambda: S.ell_spmv(rows, cols, pitch, di, dv, x, y, tuned="merge_path"))
row["ell_merge_path (fused engine)"] = {"ms": round(t, 4), "equal": bool(torch.equal(y, want))}
t, yr = ref_format(4, rows, cols, eoff, ind.reshape(-1), ev_.reshape(-1), xh)
row["ell_merge_path, reference build"] = {"ms": None if t is None else round(t, 4), "equal": None if yr is None else bool(np.array_equal(yr, want.cpu().numpy()))}
out["ELL, 2^20 rows x pitch 16"] = row; print("ELL", row, file=sys.stderr, flush=True)
del csr, di, dv

# ---- ELL with uneven rows (the case ell_merge_path exists for): C2's degrees capped at 256, pitch 256
deg = np.minimum(G.powerlaw_degrees(1 << 18, 1 << 22, cap=256), 256)
r2 = deg.size
uoff, uidx, uval = G.csr_from_degrees(deg, cols, 1)
pitch2 = int(deg.max())
ind2 = np.full((r2, pitch2), -1, np.int32); ev2 = np.zeros((r2, pitch2), np.float32)
mask = np.arange(pitch2)[None, :] < deg[:, None]
ind2[mask] = uidx; ev2[mask] = uval
csr = S.CSR.from_numpy(r2, cols, uoff, uidx, uval); want2 = S.spmv("merge_path_flat", csr, x)
di, dv = torch.from_numpy(ind2).cuda(), torch.from_numpy(ev2).cuda()
y2 = torch.empty(r2, device="cuda")
row = {"cells": int(r2 * pitch2), "nonzeros": int(uidx.size)}
for name, mode in (("tuned (row split)", True), ("reference-shaped", False), ("ell_merge_path (fused engine)", "merge_path")):
    t = ev(lambda: S.ell_spmv(r2, cols, pitch2, di, dv, x, y2, tuned=mode))
    row[name] = {"ms": round(t, 4), "equal": bool(torch.equal(y2, want2))}
for fmt, name in ((2, "reference build"), (4, "ell_merge_path, reference build")):
    t, yr = ref_format(fmt, r2, cols, uoff, uidx, uval, xh)
    row[name] = {"ms": None if t is None else round(t, 4), "equal": None if yr is None else bool(np.array_equal(yr, want2.cpu().numpy()))}
out["ELL, 2^18 power-law rows (max 256), pitch 256"] = row; print("ELL uneven", row, file=sys.stderr, flush=True)
del csr, di, dv

# ---- DIA: 2^20 rows, 33 diagonals within +-512 of the main one, 80 % filled
rngd = np.random.default_rng(1)
offs = np.unique(np.concatenate([rngd.integers(-512, 513, size=32), [0]]))
n = 1 << 20
cells = np.zeros((offs.size, n), np.float32)
rr, cc, vv = [], [], []
for k, o in enumerate(offs):
    r = np.arange(max(0, -o), min(n, n - o))
    keep = rngd.random(r.size) < 0.8
    v = (rngd.integers(1, 9, size=int(keep.sum())) / 8.0).astype(np.float32)
    cells[k, r[keep]] = v
    rr.append(r[keep]); cc.append(r[keep] + o); vv.append(v)
rr, cc, vv = np.concatenate(rr), np.concatenate(cc), np.concatenate(vv)
order = np.lexsort((cc, rr))
doff = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=n))]).astype(np.int32)
didx, dval = cc[order].astype(np.int32), vv[order]
csr = S.CSR.from_numpy(n, n, doff, didx, dval); wantd = S.spmv("merge_path_flat", csr, x)
dd, dc_ = torch.from_numpy(offs.astype(np.int32)).cuda(), torch.from_numpy(cells).cuda()
yd = torch.empty(n, device="cuda")
dia_bytes = cells.size * 4 + n * 8
row = {"diagonals": int(offs.size), "stored_cells": int(cells.size), "nonzeros": int(didx.size)}
for name, tuned in (("tuned (4 rows per lane)", True), ("reference-shaped", False)):
    t = ev(lambda: S.dia_spmv(n, n, dd, dc_, x, yd, tuned=tuned), 30)
    row[name] = {"ms": round(t, 4), "GBps": round(dia_bytes / t / 1e6, 1), "equal": bool(torch.equal(yd, wantd))}
t, yr = ref_format(3, n, n, doff, didx, dval, xh)
row["reference build"] = {"ms": None if t is None else round(t, 4), "equal": None if yr is None else bool(np.array_equal(yr, wantd.cpu().numpy()))}
t = ev(lambda: S.spmv("merge_path_flat", csr, x, yd), 30)
row["same matrix as CSR, merge_path_flat"] = {"ms": round(t, 4)}
out["DIA, 2^20 rows x 33 diagonals"] = row; print("DIA", row, file=sys.stderr, flush=True)
del csr, dd, dc_

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
